#!/bin/bash
# quick loop: range-search parity tests, then the default bench without the CPU baseline (steady state included)
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_cull.py tests/test_gpu_edge_cases.py tests/test_gpu_dev_entry_points.py tests/test_gpu_planner_loop.py -x -q -m gpu > gpurun_out/quick_pytest.log 2>&1 || { tail -40 gpurun_out/quick_pytest.log; exit 1; }
tail -2 gpurun_out/quick_pytest.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/quick_bench.json 2> gpurun_out/quick_bench.err || { tail gpurun_out/quick_bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("gpurun_out/quick_bench.json"))
print("edges/s %.4g  ms/step %.4f  steady %.4g (%.4f ms)  kernels %s" % (d["value"], d["ms_per_step"], d.get("value_steady", 0), d.get("steady_state", {}).get("ms_per_step", 0), d.get("kernel_ms")))
PY
