#!/bin/bash
# cost propagation (N4): parity tests, timings, rocprofv3 kernel stats.   usage: tools/gpu/scripts_gpu_graph.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r02_graph}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 300 python3 -m pytest tests/test_gpu_graph_cost.py -x -q > $out/pytest.log 2>&1 || { tail -30 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
timeout -k 10 200 python3 tools/bench_graph.py --out $out/graph_cost.json > /dev/null 2> $out/bench.err || { tail $out/bench.err; exit 1; }
cat $out/graph_cost.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_graph.py --cpu-nodes 2000 > $out/under_rocprof.json 2> $out/rocprof.err
echo "rocprof rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
python3 tools/kstats.py $out/kernel_stats.csv | grep -i "graph\|csr\|Name\|stats"
