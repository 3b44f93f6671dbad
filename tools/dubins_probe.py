"""Where the Dubins edge check spends its time: the same 262144 edges against 0 / 1 / 8 / 64 polygons."""
import json, math, time
import numpy as np
from rrtqx_3d_amd import synth
from rrtqx_3d_amd.context import Context
cfg = synth.CONFIGS["C3"]
pts, Q = synth.nodes(cfg.n_nodes, 4), synth.queries(cfg.batch, 4)
polys = synth.polygons(cfg.n_obstacles)
with Context(4, node_capacity=cfg.n_nodes) as ctx:
    ctx.set_wrap(3, 2 * math.pi)
    ctx.nodes_append(pts)
    off, idx, dist = ctx.nn_radius(Q[:256], 10.0, cap=1_000_000)
    owner = np.repeat(np.arange(256), np.diff(off))
    ne = min(len(idx), 262144)
    s, g = Q[owner[:ne]], pts[idx[:ne]]
    for m in (0, 1, 8, 64):
        ctx.polygons_set(polys[:m])
        ctx.dubins_edges_check(s, g, 1.0, 0.5)
        ctx.profile(2)
        for _ in range(3):
            out = ctx.dubins_edges_check(s, g, 1.0, 0.5)
        st = ctx.stats(); ctx.profile(0)
        print(json.dumps({"polygons": m, "edges": ne, "dubins_ms": round(st.ms_dubins / 3, 4), "hits": int(np.sum(out[2])) if len(out) > 2 else None}))
