#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3c
mkdir -p $out
timeout -k 10 900 python3 -m pytest tests/test_gpu_obstacle_sweep_polygon.py tests/test_gpu_obstacle_sweep.py tests/test_gpu_dubins_time.py -x -q > $out/pytest.log 2>&1
echo "rc=$?"; tail -30 $out/pytest.log
