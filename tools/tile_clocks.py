"""Where a workgroup of nn_tile_kernel spends its time: runs the fused extend() preamble at C4 on the measuring
build (python -m rrtqx_3d_amd.build --clocks) and prints, per phase boundary, the mean / median / slowest time
since the workgroup started (100 MHz wall clock, thread 0 of every workgroup).  Optional arguments: RRTX_OPT_TUNE,
then the number of 16384-node batches appended (as sorted runs) before the measured searches."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import _capi, build, synth  # noqa: E402

_capi.LIB_PATH = build.LIB_CLK
from rrtqx_3d_amd.context import Context  # noqa: E402

cfg = synth.CONFIGS["C4"]
pts, Q = synth.nodes(cfg.n_nodes, 3), synth.queries(cfg.batch, 3)
r = synth.ball_radius(cfg.n_nodes, 3)
names = ["start", "reach", "list+sample pass", "screen (wave 0)", "fence+barrier", "confirm", "hand-out", "end",
         "  wave 0 list done", "  wave 1 tail done", "  wave 2 sample done"]
with Context(3, node_capacity=cfg.n_nodes) as ctx:
    if len(sys.argv) > 1:
        ctx.set_option(_capi.RRTX_OPT_TUNE, int(sys.argv[1]))
    ctx.nodes_append(pts)
    ctx.spheres_set(synth.spheres(cfg.n_obstacles))
    n_runs = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if n_runs:
        ctx.extend_candidates(Q, r, 0.5)            # the index is built by the first search
        rng = np.random.default_rng(5)
        for _ in range(n_runs):
            ctx.nodes_append(rng.uniform(-50, 50, (cfg.batch, 3)))
        print(f"{n_runs} runs appended, {ctx.n_nodes} nodes")
    for _ in range(3):
        out = ctx.extend_candidates(Q, r, 0.5)
    L = _capi.load()
    n_wg = (cfg.batch + 15) // 16
    buf = np.zeros(n_wg * 16, dtype=np.uint64)
    L.rrtx_debug_tile_clocks.restype = C.c_int
    rc = L.rrtx_debug_tile_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(buf.size))
    assert rc == 0, rc
    clk = buf.reshape(n_wg, 16).astype(np.int64)
    t0 = clk[:, 0].min()
    rel = (clk - clk[:, :1]) / 100.0            # us since the workgroup's own start
    print(f"{n_wg} workgroups, {len(out['idx'])} neighbours; workgroup starts spread over {(clk[:, 0].max() - t0) / 100.0:.2f} us; "
          f"last end {(clk[:, 7].max() - t0) / 100.0:.2f} us after the first start")
    for k in (1, 8, 9, 10, 2, 3, 4, 5, 6, 7):
        d = rel[:, k]
        print(f"  {names[k]:18s} mean {d.mean():6.2f}  median {np.median(d):6.2f}  p95 {np.percentile(d, 95):6.2f}  max {d.max():6.2f}")
