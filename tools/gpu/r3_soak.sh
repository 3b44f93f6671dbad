#!/bin/bash
# round 3: Dubins soaks (exactness), C3 kernel split, rocprof stats of the C3 line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT"
out=gpurun_out/r3b
mkdir -p $out
timeout -k 10 300 python3 bench.py --config C3 --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_c3.json 2> $out/bench_c3.err
echo "c3 rc=$?"
python3 -c "
import json; c = json.load(open('$out/bench_c3.json')); print('C3 ms/step %.3f kernels %s' % (c['ms_per_step'], c['kernel_ms']))"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --config C3 --steps 3 --warmup 1 --no-cpu-baseline > $out/c3_rocprof.json 2> $out/c3_rocprof.err
cp $out/trace/*/*_kernel_stats.csv $out/c3_kernel_stats.csv; rm -rf $out/trace
head -8 $out/c3_kernel_stats.csv
timeout -k 10 900 python3 tools/soak_lattice.py ${1:-400} > $out/soak_lattice.log 2>&1
echo "lattice rc=$?"; tail -2 $out/soak_lattice.log
timeout -k 10 600 python3 tools/soak_dubins.py ${2:-200} > $out/soak_dubins.log 2>&1
echo "dubins rc=$?"; tail -2 $out/soak_dubins.log
