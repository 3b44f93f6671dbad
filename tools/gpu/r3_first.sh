#!/bin/bash
# round 3, first GPU pass: parity suite (Dubins now exact), short bench lines for C4 and C3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3a
mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
rc=$?
tail -15 $out/pytest_gpu.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench.json 2> $out/bench.err
echo "bench rc=$?"
timeout -k 10 300 python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
echo "c3 rc=$?"
python3 - <<PY
import json
d = json.load(open("$out/bench.json"))
print("edges/s %.4g  ms/step %.4f  kernels %s" % (d["value"], d["ms_per_step"], d["kernel_ms"]))
print("polygons:", {k: v for k, v in d["polygon_obstacles"].items() if k in ("edges_per_s", "ms_per_step")})
c = json.load(open("$out/bench_c3.json"))
print("C3 edges/s %.4g ms/step %.3f kernels %s" % (c["value"], c["ms_per_step"], c["kernel_ms"]))
PY
