#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3i
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "dubins or golden or detmath or c3_ or c5_ or edge_cases or lattice or sweep" > $out/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $out/pytest.log
timeout -k 10 300 python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
python3 -c "
import json; c = json.load(open('$out/bench_c3.json')); print('C3 ms/step %.3f kernels %s' % (c['ms_per_step'], c['kernel_ms']))"
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err
python3 - <<PY
import json
d = json.load(open("$out/bench.json"))
print("edges/s %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]))
print("host path:", {k: v for k, v in d["host_buffer_path"].items() if k != "note"})
PY
