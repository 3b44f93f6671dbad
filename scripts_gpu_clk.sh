#!/bin/bash
# quick loop for tile-kernel work: parity tests of the range search, the default bench, then the measuring build
# (-DRRTX_TILE_CLOCKS, rrtqx_3d_amd/csrc/_obj/librrtx_hip_clk.so): phase clocks per workgroup
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_slab_cull.py tests/test_gpu_edge_cases.py tests/test_gpu_dev_entry_points.py tests/test_gpu_planner_loop.py -x -q -m gpu > gpurun_out/clk_pytest.log 2>&1 || { tail -30 gpurun_out/clk_pytest.log; exit 1; }
tail -2 gpurun_out/clk_pytest.log
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/clk_bench0.json 2> gpurun_out/clk_bench0.err || exit 1
python3 tools/show_bench.py gpurun_out/clk_bench0.json 2>/dev/null | head -5
if [ -f rrtqx_3d_amd/csrc/_obj/librrtx_hip_clk.so ]; then
  cp rrtqx_3d_amd/librrtx_hip.so /tmp/librrtx_keep.so
  cp rrtqx_3d_amd/csrc/_obj/librrtx_hip_clk.so rrtqx_3d_amd/librrtx_hip.so
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err
  cp /tmp/librrtx_keep.so rrtqx_3d_amd/librrtx_hip.so
  grep "tile clocks" gpurun_out/clk_bench.err | tail -2
fi
