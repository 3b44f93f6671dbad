#!/bin/bash
# phase clocks of the tile kernel (measuring build), then the bench for each RRTX_OPT_TUNE value (no pytest)
# usage: tools/gpu/scripts_gpu_sweep2.sh <tune> [<tune> ...]
mkdir -p gpurun_out
timeout -k 10 200 python tools/tile_clocks.py > gpurun_out/tile_clocks.txt 2>&1 || { tail -5 gpurun_out/tile_clocks.txt; exit 1; }
cat gpurun_out/tile_clocks.txt
for tune in "$@"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --tune $tune > gpurun_out/bench_t$tune.json 2>> gpurun_out/bench_err.log || { tail -5 gpurun_out/bench_err.log; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_t$tune.json"))
print("tune %10d: edges/s %.4g  ms/step %.4f  nn_tile %.4f  units %s" % ($tune, d["value"], d["ms_per_step"], d["kernel_ms"]["nn_scan"], d["roofline"].get("culled_units")))
PY
done
