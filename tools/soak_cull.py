"""One-off soak: many random scenarios, culled search vs brute-force search vs exact fp64 scan
(all three through the C-ABI), plus extend_candidates culled vs unculled.  Prints a summary."""
import math, sys, time
import numpy as np
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: F401  (first: one HIP runtime image in the process)
from rrtqx_3d_amd import _capi
from rrtqx_3d_amd.context import Context

n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t0 = time.time()
tot_hits = 0
for sc in range(n_scen):
    rng = np.random.default_rng(50_000 + sc)
    d = 3 if rng.uniform() < 0.6 else 4
    n = int(rng.integers(1, 60_000))
    span = float(rng.choice([1.0, 30.0, 1e4]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        pts = rng.uniform(-span, span, (n, d))
    elif kind == 1:
        c = rng.uniform(-span, span, (int(rng.integers(1, 9)), d))
        pts = c[rng.integers(0, len(c), n)] + rng.normal(0, span / 40, (n, d))
    elif kind == 2:
        base = rng.uniform(-span, span, (max(1, n // 16), d))
        pts = base[rng.integers(0, len(base), n)]
    else:
        pts = np.round(rng.uniform(-span, span, (n, d)) / (span / 16)) * (span / 16)
    if n > 10 and rng.uniform() < 0.25:
        # a node that ruins the fp32 bounds (the screen then passes everything: slices fill up and are drained mid-screen)
        pts[int(rng.integers(1, n))] = [np.nan, 0.0, 0.0, 0.0][:d] if rng.uniform() < 0.5 else [1e9] * d
    wrap = d == 4 and rng.uniform() < 0.7
    if d == 4:
        pts[:, 3] = rng.uniform(0, 2 * math.pi, n)
    nq = int(rng.integers(1, 3000))
    Q = pts[rng.integers(0, n, nq)] + rng.normal(0, span / 20, (nq, d))
    if d == 4:
        Q[:, 3] = np.mod(Q[:, 3], 2 * math.pi)
    r = rng.uniform(0, span / 6, nq) if rng.uniform() < 0.5 else float(rng.uniform(0, span / 5))
    with Context(d, node_capacity=1024) as ctx:
        if wrap:
            ctx.set_wrap(3, 2 * math.pi)
        cuts = sorted(set([n] + [int(x) for x in rng.integers(1, n + 1, 2)]))
        done = 0
        for upto in cuts:
            ctx.nodes_append(pts[done:upto]); done = upto
            res = []
            for cull, flt in ((2, 1), (0, 1), (0, 0)):
                ctx.set_option(_capi.RRTX_OPT_NN_CULL, cull)
                ctx.set_option(_capi.RRTX_OPT_NN_FILTER, flt)
                res.append(ctx.nn_radius(Q, r))
            ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 1)
            for other in res[1:]:
                for a, b in zip(res[0], other):
                    assert np.array_equal(a, b), f"scenario {sc} n={done} d={d} kind={kind} wrap={wrap}"
            tot_hits += int(res[0][0][-1])
        if d == 3 and np.isscalar(r):
            sph = np.concatenate([rng.uniform(-span, span, (40, 3)), rng.uniform(span / 60, span / 8, (40, 1))], 1)
            ctx.spheres_set(sph)
            outs = []
            for cull in (2, 0):
                ctx.set_option(_capi.RRTX_OPT_NN_CULL, cull)
                outs.append(ctx.extend_candidates(Q, r, span / 100))
            for k in outs[0]:
                assert np.array_equal(outs[0][k], outs[1][k]), (sc, k)
    if sc % 10 == 9:
        print(f"{sc + 1} scenarios ok, {tot_hits} neighbours so far, {time.time() - t0:.0f} s", flush=True)
print("SOAK OK", n_scen, tot_hits)
