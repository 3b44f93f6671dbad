#!/bin/bash
# PMC passes for one bench line (separate passes; --kernel-trace only, as the pool requires), summarised for ONE kernel.
# usage: tools/gpu/pmc_bench.sh <tag> "<kernel name prefix>" [bench.py args ...]
#   -> gpurun_out/<tag>/pN_per_kernel_avg.csv + traffic.json (run key, instruction mix, lanes, HBM-side bytes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1; shift
kern=$1; shift
out=gpurun_out/$tag
mkdir -p $out
run() { # name counters -- bench args
  name=$1; shift; ctrs=$1; shift
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$name -- python3 bench.py --no-cpu-baseline --no-extras "$@" > $out/$name.json 2> $out/$name.err
  echo "pass $name rc=$?"
  python3 tools/pmc_summary.py $out/$name > $out/${name}_per_kernel_avg.csv
  rm -rf $out/$name
}
run p1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "$@" &&
run p2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "$@" &&
run p3 "FETCH_SIZE" "$@" &&
run p4 "WRITE_SIZE" "$@" &&
run p5 "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE" "$@" &&
python3 tools/pmc_traffic.py $out "$kern" $out/p1.json > $out/traffic.json && cat $out/traffic.json
