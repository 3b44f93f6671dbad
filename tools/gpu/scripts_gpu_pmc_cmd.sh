#!/bin/bash
# Two PMC passes (instruction mix, lane utilisation) for an arbitrary python script of this repo.
# usage: tools/gpu/scripts_gpu_pmc_cmd.sh <tag> <script.py> [args]   -> gpurun_out/<tag>/pN_per_kernel_avg.csv
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT"
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
run() {
  name=$1; shift; ctrs=$1; shift
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$name -- python3 "$@" > $out/$name.log 2> $out/$name.err
  echo "pass $name rc=$?"
  python3 tools/pmc_summary.py $out/$name > $out/${name}_per_kernel_avg.csv
  rm -rf $out/$name
}
run p1 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD" "$@" &&
run p2 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU" "$@"
