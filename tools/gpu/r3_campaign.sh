#!/bin/bash
# round-3 measurement campaign: parity suite, the four bench lines (C4 spheres = the driver's line, C4 polygons, C3, C5)
# with CPU baselines, rocprofv3 kernel stats of the same commands, PMC passes keyed by run, 2-rank rehearsals, soaks.
# usage: tools/gpu/r3_campaign.sh <tag> [part ...]   parts: tests bench stats pmc ranks soaks (default: all)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
tag=${1:-r03_v2}; shift
parts=${*:-tests bench stats pmc ranks soaks}
out=gpurun_out/$tag
mkdir -p $out
has() { [[ " $parts " == *" $1 "* ]]; }
if has tests; then
  timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1
  rc=$?; tail -3 $out/pytest_gpu.log; echo "pytest rc=$rc"; [ $rc -ne 0 ] && exit $rc
fi
if has bench; then
  timeout -k 10 500 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"
  timeout -k 10 500 python3 bench.py --obstacles polygons > $out/bench_poly.json 2> $out/bench_poly.err; echo "bench_poly rc=$?"
  timeout -k 10 500 python3 bench.py --config C3 --steps 5 --warmup 1 > $out/bench_c3.json 2> $out/bench_c3.err; echo "bench_c3 rc=$?"
  timeout -k 10 700 python3 bench.py --config C5 --steps 6 --warmup 2 > $out/bench_c5.json 2> $out/bench_c5.err; echo "bench_c5 rc=$?"
fi
if has stats; then
  for v in "bench:" "bench_static:--no-extras" "bench_poly:--obstacles polygons --no-extras" "bench_c3:--config C3 --steps 3 --warmup 1" "bench_c5:--config C5 --steps 2 --warmup 1"; do
    name=${v%%:*}; args=${v#*:}
    timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 bench.py --no-cpu-baseline $args > $out/${name}_under_rocprof.json 2> $out/${name}_rocprof.err
    echo "rocprof $name rc=$?"
    cp $out/trace_$name/*/*_kernel_stats.csv $out/kernel_stats_$name.csv 2>/dev/null; rm -rf $out/trace_$name
  done
fi
if has pmc; then
  bash tools/gpu/pmc_bench.sh $tag/pmc_c4 "nn_tile_kernel<3, true>" --steps 5 --warmup 2 > $out/pmc_c4.log 2>&1; echo "pmc c4 rc=$?"
  bash tools/gpu/pmc_bench.sh $tag/pmc_poly "edges_polygons_kernel" --obstacles polygons --steps 5 --warmup 2 > $out/pmc_poly.log 2>&1; echo "pmc poly rc=$?"
  python3 tools/pmc_traffic.py $out/pmc_poly "points_polygons_flag_kernel" $out/pmc_poly/p1.json > $out/pmc_poly/traffic_points.json
  python3 tools/pmc_traffic.py $out/pmc_poly "nn_tile_kernel<3, false>" $out/pmc_poly/p1.json > $out/pmc_poly/traffic_tile.json
  bash tools/gpu/pmc_bench.sh $tag/pmc_c3 "dubins_check_rec_kernel<false>" --config C3 --steps 2 --warmup 1 > $out/pmc_c3.log 2>&1; echo "pmc c3 rc=$?"
  python3 tools/pmc_traffic.py $out/pmc_c3 "dubins_steer_rec_kernel" $out/pmc_c3/p1.json > $out/pmc_c3/traffic_steer.json
  bash tools/gpu/pmc_bench.sh $tag/pmc_c5 "dubins_check_rec_kernel<true>" --config C5 --steps 2 --warmup 1 > $out/pmc_c5.log 2>&1; echo "pmc c5 rc=$?"
fi
if has ranks; then
  for v in "2rank_weak:" "2rank_strong:--scaling strong" "2rank_obstacles:--shard obstacles --obstacles polygons" "4rank_grid2x2:--grid 2x2 --obstacles polygons"; do
    name=${v%%:*}; args=${v#*:}; n=${name%%rank*}
    timeout -k 10 400 python3 bench.py --gpus $n --backend gloo --share-device --no-cpu-baseline --no-extras --steps 8 --warmup 2 $args > $out/bench_$name.json 2> $out/bench_$name.err
    echo "ranks $name rc=$?"
  done
fi
if has soaks; then
  timeout -k 10 900 python3 tools/soak_lattice.py 400 > $out/soak_lattice.log 2>&1; echo "lattice rc=$?"; tail -1 $out/soak_lattice.log
  timeout -k 10 600 python3 tools/soak_dubins.py 200 > $out/soak_dubins.log 2>&1; echo "dubins rc=$?"; tail -1 $out/soak_dubins.log
  timeout -k 10 600 python3 tools/soak_polygons.py 300 > $out/soak_polygons.log 2>&1; echo "polygons rc=$?"; tail -1 $out/soak_polygons.log
  timeout -k 10 900 python3 tools/soak_cull.py 100 > $out/soak_cull.log 2>&1; echo "cull rc=$?"; tail -1 $out/soak_cull.log
fi
python3 - <<PY
import json, os
o = "$out"
def ld(n):
    try: return json.loads(open(os.path.join(o, n)).read().strip().splitlines()[-1])
    except Exception as e: return None
d = ld("bench.json")
if d:
    print("C4 edges/s %.4g ms/step %.4f steady %.4g cpu %.4g frac %.3f" % (d["value"], d["ms_per_step"], d.get("value_steady", 0), d.get("cpu_baseline", {}).get("value", 0), d["roofline"]["frac"]))
    print("  host", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["host_buffer_path"].items() if k in ("ms_per_step",)}, "poly", d["polygon_obstacles"]["ms_per_step"], "large", d["large_batch"]["value"])
for n in ("bench_poly.json", "bench_c3.json", "bench_c5.json", "bench_2rank_weak.json", "bench_2rank_strong.json", "bench_2rank_obstacles.json", "bench_4rank_grid2x2.json"):
    d = ld(n)
    if d: print(n, "n_gpus", d["n_gpus"], "edges/s %.4g ms/step %.4f" % (d["value"], d["ms_per_step"]), "cpu", d.get("cpu_baseline", {}).get("value"), "frac", d["roofline"].get("frac"))
PY
