"""Probe of the slab-culled range scan on one config: unit count and kernel-family times."""
import json, sys, time
import numpy as np
from rrtqx_3d_amd import synth, _capi
from rrtqx_3d_amd.context import Context

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
N, B = cfg.n_nodes, cfg.batch
pts, Q = synth.nodes(N, cfg.dim), synth.queries(B, cfg.dim)
r = synth.ball_radius(N, 3) if cfg.dim == 3 else 10.0
with Context(cfg.dim) as ctx:
    if cfg.dim == 4:
        ctx.set_wrap(3, 2 * np.pi)
    ctx.nodes_append(pts)
    for mode in (2, 0):
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, mode)
        for tq in (64, 32, 16):
            ctx.set_option(_capi.RRTX_OPT_SCAN_TILE_Q, tq)
            ctx.nn_radius(Q, r)
            ctx.profile(2)
            t0 = time.perf_counter()
            for _ in range(5):
                off, idx, dist = ctx.nn_radius(Q, r)
            wall = (time.perf_counter() - t0) / 5
            st = ctx.stats()
            ctx.profile(0)
            print(json.dumps({"cull": mode, "tile_q": tq, "units": st.last_scan_units, "neighbors": int(off[-1]),
                              "scan_ms": st.ms_nn_scan / max(1, st.launches_nn_scan),
                              "finish_ms": st.ms_nn_finish / max(1, st.launches_nn_scan), "wall_ms": wall * 1e3}))
