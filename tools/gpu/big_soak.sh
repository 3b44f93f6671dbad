#!/bin/bash
# the four soaks at five times the campaign's size (tools/gpu/r3_campaign.sh runs 400 / 200 / 300 / 100 scenes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT" HSA_ENABLE_IPC_MODE_LEGACY=0
out=gpurun_out/${1:-big_soak}; mkdir -p $out
timeout -k 10 1000 python3 tools/soak_lattice.py 2000 > $out/soak_lattice_2000.log 2>&1; rc1=$?; echo "lattice rc=$rc1"; tail -1 $out/soak_lattice_2000.log
timeout -k 10 600 python3 tools/soak_dubins.py 1000 > $out/soak_dubins_1000.log 2>&1; rc2=$?; echo "dubins rc=$rc2"; tail -1 $out/soak_dubins_1000.log
timeout -k 10 600 python3 tools/soak_polygons.py 1500 > $out/soak_polygons_1500.log 2>&1; rc3=$?; echo "polygons rc=$rc3"; tail -1 $out/soak_polygons_1500.log
timeout -k 10 900 python3 tools/soak_cull.py 400 > $out/soak_cull_400.log 2>&1; rc4=$?; echo "cull rc=$rc4"; tail -1 $out/soak_cull_400.log
exit $((rc1 + rc2 + rc3 + rc4))
