#!/bin/bash
# kernel stats of the timed line alone (static index: no steady-state / polygon / host-path passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-r02_static}
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
echo "rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
python3 tools/kstats.py $out/kernel_stats.csv | grep "nn_"
python3 tools/show_bench.py $out/bench_under_rocprof.json
