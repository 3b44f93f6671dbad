#!/bin/bash
# standard GPU check: parity tests, then a short bench; prints the kernel split
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > gpurun_out/bench_tmp.json 2>> gpurun_out/bench_err.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_tmp.json"))
print("edges/s %.4g  ms/step %.4f  kernels %s  hbm_frac %.3f" % (d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["kernel_ms"].items()}, d["roofline"]["frac"]))
PY
