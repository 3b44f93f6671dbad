#!/bin/bash
# bench under rocprofv3 --kernel-trace --stats for each RRTX_OPT_TUNE value (no parity tests: diagnostic bits may break results)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
for tune in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tune -- python3 bench.py --no-cpu-baseline --tune $tune > $out/bench_t$tune.json 2> $out/rocprof_t$tune.err || { tail -5 $out/rocprof_t$tune.err; exit 1; }
  cp $out/trace_$tune/*/*_kernel_stats.csv $out/kernel_stats_t$tune.csv && rm -rf $out/trace_$tune
  echo "tune $tune done"
done
