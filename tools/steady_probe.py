"""How the cost of one fused extend() step grows with the number of sorted runs appended since the last rebuild
(GPU box).  Prints per step: nodes, chunks screened per tile, ms of the step's search."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401

from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

N, B, STEPS = 200_000, 16384, 16
pts = synth.nodes(N + B * STEPS, 3)
sph = synth.spheres(256) if hasattr(synth, "spheres") else None
with Context(3, node_capacity=N + B * (STEPS + 1)) as ctx:
    if sph is not None:
        ctx.spheres_set(sph, np.ones(len(sph), dtype=np.uint8))
    ctx.nodes_append(pts[:N])
    rng = np.random.default_rng(1)
    for s in range(STEPS):
        n = ctx.n_nodes
        r = synth.ball_radius(n, 3)
        Q = rng.uniform(-50, 50, (B, 3))
        ctx.extend_candidates(Q, r, 0.5)                       # warm (and possibly a rebuild)
        t0 = time.perf_counter()
        for _ in range(3):
            out = ctx.extend_candidates(Q, r, 0.5)
        dt = (time.perf_counter() - t0) / 3
        st = ctx.stats()
        print("step %2d n %7d r %.3f  chunks/tile %6.2f  host-path ms %.3f  neighbours %d" % (
            s, n, r, st.last_scan_units / (B / 16), 1e3 * dt, len(out["idx"]) if "idx" in out else -1), flush=True)
        ctx.nodes_append(pts[N + s * B:N + (s + 1) * B])
