#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r3w; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "polygon or lattice or full_size or edge_cases or planner or moving or sweep" > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/polygon_clocks.py > $out/polygon_clocks.txt 2>&1; echo "rc=$?"; head -11 $out/polygon_clocks.txt
timeout -k 10 300 python3 tools/polygon_cap_probe.py > $out/cap_probe.txt 2>&1; echo "rc=$?"; tail -2 $out/cap_probe.txt
timeout -k 10 300 python3 bench.py --obstacles polygons --no-cpu-baseline --no-extras > $out/bench_poly.json 2> $out/bench_poly.err; echo "rc=$?"
python3 -c "
import json; d=json.loads(open('$out/bench_poly.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
