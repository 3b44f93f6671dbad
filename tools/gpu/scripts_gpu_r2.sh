#!/bin/bash
# round-2 GPU check: parity tests, bench line, rocprofv3 kernel stats of the same bench command
# usage: tools/gpu/scripts_gpu_r2.sh <tag> [pytest-args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r02_a}; shift
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q "$@" > $out/pytest_gpu.log 2>&1
rc=$?
tail -15 $out/pytest_gpu.log
echo "pytest rc=$rc"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 - <<PY
import json
d = json.load(open("$out/bench.json"))
print("edges/s %.4g  ms/step %.4f  kernels %s  frac %.3f" % (d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["kernel_ms"].items()}, d["roofline"]["frac"]))
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
echo "rocprof rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv && rm -rf $out/trace
head -14 $out/kernel_stats.csv | cut -c1-150
