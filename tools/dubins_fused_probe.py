"""rrtx_extend_candidates_dubins at BASELINE config C3 (N = 50 k, 64 polygons, B = 4096, r = 10), two
calls: meant to be run under rocprofv3 (kernel trace or --pmc)."""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

cfg = synth.CONFIGS["C3"]
pts, Q = synth.nodes(cfg.n_nodes, 4), synth.queries(cfg.batch, 4)
with Context(4, node_capacity=cfg.n_nodes) as ctx:
    ctx.set_wrap(3, 2 * math.pi)
    ctx.nodes_append(pts)
    ctx.polygons_set(synth.polygons(cfg.n_obstacles))
    for _ in range(2):
        out = ctx.extend_candidates_dubins(Q, 10.0, 0.5, 1.0, cap=6_000_000)
    print(len(out["idx"]), float(out["hit_out"].mean()))
