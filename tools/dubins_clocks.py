"""Where the waves of the fused Dubins preamble spend their time (config C3): runs rrtx_extend_candidates_dubins on
the measuring build (python -m rrtqx_3d_amd.build --clocks) and prints the share of every stage of
candidate_dubins_kernel in the waves' summed wall-clock ticks."""
import ctypes as C
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import _capi, build, synth  # noqa: E402

_capi.LIB_PATH = build.LIB_CLK
from rrtqx_3d_amd.context import Context  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = synth.CONFIGS[which]
if which == "C5":          # Dubins edges with time, a quarter of the polygons moving (python tools/dubins_clocks.py C5)
    pts, Q = synth.nodes_time(cfg.n_nodes), synth.nodes_time(1024, seed=synth.SEED + 1)
    polys, kinds, paths, active, hidden = synth.dynamic_polygons(cfg.n_obstacles)
    r_q, r_min_q, cap_q = synth.ball_radius(cfg.n_nodes, 4, gamma=100.0, delta=10.0), synth.R_MIN_TIME, 3_000_000
else:
    pts, Q = synth.nodes(cfg.n_nodes, 4), synth.queries(cfg.batch, 4)
    polys, kinds, paths, active = synth.polygons(cfg.n_obstacles), None, None, None
    r_q, r_min_q, cap_q = 10.0, 1.0, 6_000_000
names = ["steering (both directions)", "stage 1a boxes", "stage 1b chord tests", "arc screen", "stage 2a pieces / circles",
         "stage 2b polygon tests", "waves"]
with Context(4, node_capacity=cfg.n_nodes) as ctx:
    ctx.set_wrap(3, 2 * math.pi)
    ctx.nodes_append(pts)
    ctx.polygons_set(polys, kinds=kinds, paths=paths, active=active)
    if which == "C5":
        ctx.set_space_has_time(True)
        ctx.set_dubins_velocity(synth.V_MIN, synth.V_MAX)
    L = _capi.load()
    L.rrtx_debug_dubins_clocks.restype = C.c_int
    buf = np.zeros(8, dtype=np.uint64)
    out = ctx.extend_candidates_dubins(Q, r_q, 0.5, r_min_q, cap=cap_q)          # warm
    assert L.rrtx_debug_dubins_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    out = ctx.extend_candidates_dubins(Q, r_q, 0.5, r_min_q, cap=cap_q)
    assert L.rrtx_debug_dubins_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(0)) == 0
    t = buf[:6].astype(np.float64)
    print(f"{which}: {len(out['idx'])} neighbours, {int(buf[6])} waves, hit fraction {float((out['hit_out'] & 1).mean()):.3f}")
    for k in range(6):
        print(f"  {names[k]:28s} {100.0 * t[k] / t.sum():5.1f} %   ({t[k] / 100.0 / max(int(buf[6]), 1):7.2f} us per wave)")
