"""The fused extend() preamble against the polygon list at C4 scale (edges from the CSR lists: edges_polygons_kernel
in CSR mode, points_polygons_kernel without a certificate), a few calls: meant to be run under rocprofv3
(kernel trace or --pmc, tools/gpu/scripts_gpu_pmc_cmd.sh)."""
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
    for _ in range(4):
        out = ctx.extend_candidates(Q, synth.ball_radius(cfg.n_nodes, 3), 0.5)
    print(len(out["idx"]), float(out["hit_out"].mean()), float(out["sample_unsafe"].mean()))
