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

cfg = synth.CONFIGS["C3"]
pts, Q = synth.nodes(cfg.n_nodes, 4), synth.queries(cfg.batch, 4)
names = ["steering (both directions)", "stage 1a boxes", "stage 1b chord tests", "arc screen", "stage 2a pieces / circles",
         "stage 2b polygon tests", "waves"]
with Context(4, node_capacity=cfg.n_nodes) as ctx:
    ctx.set_wrap(3, 2 * math.pi)
    ctx.nodes_append(pts)
    ctx.polygons_set(synth.polygons(cfg.n_obstacles))
    L = _capi.load()
    L.rrtx_debug_dubins_clocks.restype = C.c_int
    buf = np.zeros(8, dtype=np.uint64)
    out = ctx.extend_candidates_dubins(Q, 10.0, 0.5, 1.0, cap=6_000_000)          # warm
    assert L.rrtx_debug_dubins_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(1)) == 0
    out = ctx.extend_candidates_dubins(Q, 10.0, 0.5, 1.0, cap=6_000_000)
    assert L.rrtx_debug_dubins_clocks(buf.ctypes.data_as(C.c_void_p), C.c_int(0)) == 0
    t = buf[:6].astype(np.float64)
    print(f"{len(out['idx'])} neighbours, {int(buf[6])} waves, hit fraction {float(out['hit_out'].mean()):.3f}")
    for k in range(6):
        print(f"  {names[k]:28s} {100.0 * t[k] / t.sum():5.1f} %   ({t[k] / 100.0 / max(int(buf[6]), 1):7.2f} us per wave)")
