#!/bin/bash
# tuning sweep of the range-scan launch geometry (prints one line per point)
mkdir -p gpurun_out
for sb in 1280 2048 2560 4096; do for it in 2048 4096 8192; do for tq in 32 64 128; do
timeout -k 10 120 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --scan-blocks $sb --scan-items $it --tile-q $tq > gpurun_out/sweep.json 2>> gpurun_out/bench_err.log
python - "$sb" "$it" "$tq" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sweep.json"))
print("blocks", sys.argv[1], "items", sys.argv[2], "tile_q", sys.argv[3], "scan_ms %.4f step_ms %.4f" % (d["kernel_ms"]["nn_scan"], d["ms_per_step"]))
PY
done; done; done
