#!/bin/bash
# PMC passes for the bench (separate passes; --kernel-trace only, as the pool requires)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_r1
mkdir -p $out
rocprofv3 -L > $out/counters_list.txt 2>&1
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $out/$name -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/$name.json 2> $out/$name.err
  echo "pass $name rc=$?"
}
run p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU
run p3 FETCH_SIZE
run p4 WRITE_SIZE
run p5 TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
ls -R $out | head -40
