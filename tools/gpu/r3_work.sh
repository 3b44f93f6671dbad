#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r3w; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dubins or lattice or full_size or sweep or time" > $out/pytest.log 2>&1; rc=$?; tail -3 $out/pytest.log
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $out/c3.json 2> $out/c3.err; echo "c3 rc=$?"
python3 -c "
import json; d=json.loads(open('$out/c3.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('kernel_ms'))"
timeout -k 10 300 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $out/c5.json 2> $out/c5.err; echo "c5 rc=$?"
python3 -c "
import json; d=json.loads(open('$out/c5.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('phase_ms'))"
timeout -k 10 400 python3 tools/soak_dubins.py 60 > $out/soak_dubins.log 2>&1; echo "soak rc=$?"; tail -1 $out/soak_dubins.log
exit 0
