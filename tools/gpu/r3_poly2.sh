#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT"
out=gpurun_out/r3g
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "polygon or lattice or golden or moving or c2_ or c4_ or edge_cases or sweep or kat" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $out/pytest.log
timeout -k 10 900 python3 tools/soak_lattice.py ${1:-400} > $out/soak_lattice.log 2>&1
echo "lattice rc=$?"; tail -2 $out/soak_lattice.log
timeout -k 10 600 python3 tools/soak_polygons.py 100 > $out/soak_polygons.log 2>&1
echo "soak_polygons rc=$?"; tail -2 $out/soak_polygons.log
timeout -k 10 300 python3 bench.py --obstacles polygons --no-cpu-baseline > $out/bench_poly.json 2> $out/bench_poly.err
python3 - <<PY
import json
d = json.load(open("$out/bench_poly.json"))
print("poly edges/s %.4g ms/step %.4f kernels %s" % (d["value"], d["ms_per_step"], d["kernel_ms"]))
PY
