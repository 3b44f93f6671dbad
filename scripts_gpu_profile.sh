#!/bin/bash
# round profile: default bench (with CPU baseline) + rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r01_v2}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
echo "rocprof rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
cat $out/bench.json
