#!/bin/bash
# round 3: bench lines (default, polygons, C5 short) + 2-rank rehearsal of the obstacle-shard / grid paths on one GPU
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3d
mkdir -p $out
run() { name=$1; shift; timeout -k 10 ${TMO:-400} "$@" > $out/$name.json 2> $out/$name.err; echo "$name rc=$?"; tail -c 600 $out/$name.err; }
TMO=500 run c5_short python3 bench.py --config C5 --steps 3 --warmup 1
python3 - <<PY
import json
try:
    c = json.load(open("$out/c5_short.json"))
    print("C5 edges/s %.4g ms/cycle %.1f phases %s per_cycle %s cpu %s" % (c["value"], c["ms_per_step"], {k: round(v, 2) for k, v in c["phase_ms"].items()}, c["per_cycle"], c.get("cpu_baseline", {}).get("value")))
    print(c["kernel_ms"], c["config"]["edge_mirror"])
except Exception as e: print("c5 parse", e)
PY
run bench python3 bench.py --no-cpu-baseline
run bench_poly python3 bench.py --obstacles polygons --no-cpu-baseline
python3 - <<PY
import json
for n in ("bench", "bench_poly"):
    try:
        d = json.load(open("$out/%s.json" % n))
        print(n, "edges/s %.4g ms/step %.4f kernels %s large %s" % (d["value"], d["ms_per_step"], d["kernel_ms"], d.get("large_batch")))
        if "polygon_obstacles" in d: print("  polygons:", d["polygon_obstacles"]["ms_per_step"], d["polygon_obstacles"]["kernel_ms"])
    except Exception as e: print(n, "parse", e)
PY
export HSA_ENABLE_IPC_MODE_LEGACY=0
run grid_obs python3 bench.py --gpus 2 --backend gloo --share-device --shard obstacles --obstacles polygons --no-cpu-baseline --no-extras --steps 8 --warmup 2
run grid_2x1 python3 bench.py --gpus 2 --backend gloo --share-device --no-cpu-baseline --no-extras --steps 8 --warmup 2
python3 - <<PY
import json
for n in ("grid_obs", "grid_2x1"):
    try:
        d = json.load(open("$out/%s.json" % n))
        print(n, "n_gpus", d["n_gpus"], "edges/s %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), d["config"]["grid"], d["config"]["collective"][:80])
    except Exception as e: print(n, "parse", e)
PY
