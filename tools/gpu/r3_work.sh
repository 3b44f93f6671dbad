#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r3w; mkdir -p $out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "grid_follows or flag_without" > $out/pytest.log 2>&1; rc=$?; tail -8 $out/pytest.log
exit $rc
