#!/bin/bash
# PMC passes for the C3 line (fused Dubins preamble): instruction mix and lane utilisation of candidate_dubins_kernel.
# usage: tools/gpu/scripts_gpu_pmc_c3.sh <tag>   -> gpurun_out/<tag>/pN_per_kernel_avg.csv + lanes.json
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r02_pmc_dubins_fused}
out=gpurun_out/$tag
mkdir -p $out
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 bench.py --config C3 --steps 2 --warmup 1 --no-cpu-baseline > $out/$name.json 2> $out/$name.err
  echo "pass $name rc=$?"
  python3 tools/pmc_summary.py $out/$name > $out/${name}_per_kernel_avg.csv
  rm -rf $out/$name
}
run p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
python3 - <<PY > $out/lanes.json
import csv, json
def get(f, kern, c):
    for r in csv.DictReader(open("$out/" + f)):
        if r["counter"] == c and kern in r["kernel"]:
            return float(r["avg_per_launch"])
k = "candidate_dubins_kernel"
valu = get("p1_per_kernel_avg.csv", k, "SQ_INSTS_VALU")
act = get("p2_per_kernel_avg.csv", k, "SQ_ACTIVE_INST_VALU")
thr = get("p2_per_kernel_avg.csv", k, "SQ_THREAD_CYCLES_VALU")
wait = get("p2_per_kernel_avg.csv", k, "SQ_WAIT_ANY")
wc = get("p1_per_kernel_avg.csv", k, "SQ_WAVE_CYCLES")
print(json.dumps({"kernel": k, "config": "C3", "SQ_INSTS_VALU": valu, "SQ_INSTS_SALU": get("p1_per_kernel_avg.csv", k, "SQ_INSTS_SALU"),
                  "SQ_ACTIVE_INST_VALU": act, "SQ_THREAD_CYCLES_VALU": thr,
                  "active_lanes_per_valu_instruction": (thr / act if act else None),
                  "note": "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU = average number of active lanes of a VALU instruction (of 64)",
                  "SQ_WAIT_ANY_over_SQ_WAVE_CYCLES": (wait / wc if wc else None)}))
PY
cat $out/lanes.json
