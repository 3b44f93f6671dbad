#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3h
mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - <<PY
import json
d = json.load(open("$out/bench.json"))
print("edges/s %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]))
print("host path:", d["host_buffer_path"])
print("poly:", d["polygon_obstacles"]["ms_per_step"], d["polygon_obstacles"]["kernel_ms"])
PY
