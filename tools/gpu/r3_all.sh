#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3j
mkdir -p $out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $out/pytest.log
TMO=500 timeout -k 10 500 python3 bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $out/c5.json 2> $out/c5.err
python3 - <<PY
import json
c = json.load(open("$out/c5.json"))
print("C5 edges/s %.4g ms/cycle %.1f phases %s" % (c["value"], c["ms_per_step"], {k: round(v, 2) for k, v in c["phase_ms"].items()}))
print(c["kernel_ms"])
PY
