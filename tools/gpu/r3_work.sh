#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/r3w; mkdir -p $out
timeout -k 10 300 python3 tools/polygon_clocks.py > $out/polygon_clocks.txt 2>&1; echo "rc=$?"; head -12 $out/polygon_clocks.txt
exit 0
