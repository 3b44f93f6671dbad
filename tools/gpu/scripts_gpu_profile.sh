#!/bin/bash
# round profile: parity tests, default bench (with CPU baseline) + rocprofv3 kernel stats of the same command,
# the C3 line.   usage: tools/gpu/scripts_gpu_profile.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r02_v1}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
rc=$?
tail -4 $out/pytest_gpu.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
echo "rocprof rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
timeout -k 10 300 python3 bench.py --config C3 --steps 5 --warmup 1 > $out/bench_c3.json 2> $out/bench_c3.err
echo "c3 rc=$?"
python3 - <<PY
import json
d = json.load(open("$out/bench.json"))
print("edges/s %.4g  ms/step %.4f  steady %.4g (%.4f ms)  frac %.3f  cpu %.4g" % (d["value"], d["ms_per_step"], d.get("value_steady", 0), d.get("steady_state", {}).get("ms_per_step", 0), d["roofline"]["frac"], d["cpu_baseline"]["value"]))
print("polygons:", {k: v for k, v in d["polygon_obstacles"].items() if k in ("edges_per_s", "ms_per_step")})
c = json.load(open("$out/bench_c3.json"))
print("C3 edges/s %.4g ms/step %.3f kernels %s cpu %s" % (c["value"], c["ms_per_step"], c["kernel_ms"], c.get("cpu_baseline", {}).get("value")))
PY
