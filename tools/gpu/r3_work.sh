#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r3w; mkdir -p $out
for t in 0 16 32 64 0 16 32 64; do
timeout -k 10 300 python3 bench.py --obstacles polygons --no-cpu-baseline --no-extras --tune $t > $out/bench_poly_$t.json 2> $out/bench_poly_$t.err; echo "tune $t rc=$?"
python3 -c "
import json; d=json.loads(open('$out/bench_poly_$t.json').read().strip().splitlines()[-1]); print($t, d['value'], d['ms_per_step'])"
done
