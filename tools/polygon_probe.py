"""Straight-edge polygon kernel at C4 scale (the candidate edges of one bench step against 256 random
polygons), a few calls: meant to be run under rocprofv3 (kernel trace or --pmc)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

cfg = synth.CONFIGS["C4"]
pts, Q = synth.nodes(cfg.n_nodes, 3), synth.queries(cfg.batch, 3)
with Context(3, node_capacity=cfg.n_nodes) as ctx:
    ctx.nodes_append(pts)
    off, idx, _ = ctx.nn_radius(Q, synth.ball_radius(cfg.n_nodes, 3))
    p0, p1 = synth.candidate_edges(Q, pts, off, idx)
    ctx.polygons_set(synth.polygons(cfg.n_obstacles))
    for _ in range(4):
        hit, first = ctx.edges_check(p0, p1, 0.5, kind=1)
    print(len(p0), float(hit.mean()))
