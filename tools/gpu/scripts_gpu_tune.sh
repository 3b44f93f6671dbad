#!/bin/bash
# parity tests once, then the bench under rocprofv3 --kernel-trace --stats for each RRTX_OPT_TUNE value
# usage: tools/gpu/scripts_gpu_tune.sh <tag> <tune> [<tune> ...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
rc=$?
tail -4 $out/pytest_gpu.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
for tune in "$@"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$tune -- python3 bench.py --no-cpu-baseline --tune $tune > $out/bench_t$tune.json 2> $out/rocprof_t$tune.err || { tail -5 $out/rocprof_t$tune.err; exit 1; }
  cp $out/trace_$tune/*/*_kernel_stats.csv $out/kernel_stats_t$tune.csv && rm -rf $out/trace_$tune
  python3 - <<PY
import json, csv
d = json.load(open("$out/bench_t$tune.json"))
print("tune $tune: edges/s %.4g  ms/step %.4f (under rocprof)" % (d["value"], d["ms_per_step"]))
for r in csv.reader(open("$out/kernel_stats_t$tune.csv")):
    if r[0] == "Name" or "rrtx" not in r[0]: continue
    n = r[0].split("::")[-1].split("(")[0]
    if any(k in n for k in ("nn_tile", "nn_finish", "nn_pack", "nn_place", "sample_sph", "candidate_edges")):
        print("   %-40s calls %4s avg %7.2f us" % (n[:40], r[1], float(r[3]) / 1000))
PY
done
