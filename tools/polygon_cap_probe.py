"""How much of edges_polygons_kernel's time is the grid over the caller's capacity: the fused extend step against 256
polygons (C4) with the CSR capacity at 96, 48 and 28 entries per sample (the lists hold about 25)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import _capi, synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

cfg = synth.CONFIGS["C4"]
pts, Q = synth.nodes(cfg.n_nodes, 3), synth.queries(cfg.batch, 3)
with Context(3, node_capacity=cfg.n_nodes) as ctx:
    ctx.nodes_append(pts)
    ctx.polygons_set(synth.polygons(cfg.n_obstacles))
    ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
    r = synth.ball_radius(cfg.n_nodes, 3)
    ctx.extend_candidates(Q, r, 0.5, cap=96 * cfg.batch)
    ctx.profile(2)
    for per in (96, 48, 28, 96):
        s0 = ctx.stats()
        e0, l0, p0 = s0.ms_edges, s0.launches_edges, s0.ms_points
        for _ in range(10):
            out = ctx.extend_candidates(Q, r, 0.5, cap=per * cfg.batch)
        s1 = ctx.stats()
        print("cap %d/sample: %d entries, edges kernel %.4f ms, points %.4f ms" % (per, len(out["idx"]), (s1.ms_edges - e0) / max(1, s1.launches_edges - l0), (s1.ms_points - p0) / 10))
