#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT"
out=gpurun_out/r3l
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dubins or golden or c3_ or c5_ or lattice or sweep or shards" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $out/pytest.log
timeout -k 10 300 python3 tools/soak_dubins.py 60 > $out/soak_dubins.log 2>&1; echo "soak rc=$?"; tail -1 $out/soak_dubins.log
timeout -k 10 300 python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
python3 -c "
import json; c = json.load(open('$out/bench_c3.json')); print('C3 ms/step %.3f kernels %s' % (c['ms_per_step'], c['kernel_ms']))"
timeout -k 10 500 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $out/c5.json 2> $out/c5.err
python3 -c "
import json; c = json.load(open('$out/c5.json')); print('C5 ms/cycle %.1f preamble %.1f' % (c['ms_per_step'], c['phase_ms']['extend_preamble']), c['kernel_ms'])"
