"""A few culled range searches on config C4 for rocprofv3 --kernel-trace --stats."""
import sys
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
    ctx.set_option(_capi.RRTX_OPT_NN_CULL, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    if len(sys.argv) > 3:
        ctx.set_option(_capi.RRTX_OPT_SCAN_TILE_Q, int(sys.argv[3]))
    for _ in range(20):
        ctx.nn_radius(Q, r)
