"""One-off soak: rrtx_nn_knearest, list path vs exhaustive kernel (bit for bit) on random scenes, with
oracle spot checks (kd-tree + max-heap restatement of the reference)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from rrtqx_3d_amd import _capi  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.time()
for sc in range(n_scen):
    rng = np.random.default_rng(70_000 + sc)
    d = 3 if rng.uniform() < 0.7 else 4
    n = int(rng.integers(8192, 90_000))
    span = float(rng.choice([1.0, 50.0, 1e4]))
    kind = int(rng.integers(0, 4))
    if kind == 0:
        pts = rng.uniform(-span, span, (n, d))
    elif kind == 1:
        c = rng.uniform(-span, span, (int(rng.integers(1, 9)), d))
        pts = c[rng.integers(0, len(c), n)] + rng.normal(0, span / 40, (n, d))
    elif kind == 2:
        base = rng.uniform(-span, span, (max(1, n // 16), d))
        pts = base[rng.integers(0, len(base), n)]               # 16-fold duplicates: ties everywhere
    else:
        pts = np.round(rng.uniform(-span, span, (n, d)) / (span / 16)) * (span / 16)
    nq = int(rng.integers(256, 2500))
    Q = pts[rng.integers(0, n, nq)] + rng.normal(0, span / 20, (nq, d))
    Q[:3] = span * 50                                            # far outside
    k = int(rng.choice([1, 2, 7, 16, 64, 128, 200, 512]))
    with Context(d) as ctx:
        ctx.nodes_append(pts)
        ctx.set_option(_capi.RRTX_OPT_KNN_LISTS, 1)
        a = ctx.nn_knearest(Q, k)
        ctx.set_option(_capi.RRTX_OPT_KNN_LISTS, 0)
        b = ctx.nn_knearest(Q, k)
    for x, y in zip(a, b):
        assert np.array_equal(x, y), f"scenario {sc}: list path and exhaustive kernel differ"
    if kind in (0, 1):                                           # tie-free scenes: the oracle's set is unique
        t = O.KDTree(d)
        t.insert_many(pts)
        for i in rng.choice(nq, 12, replace=False):
            oi, ok = t.knearest(k, Q[i])
            o, g = np.argsort(oi), np.argsort(a[0][i])
            assert np.array_equal(a[0][i][g], oi[o]) and np.array_equal(a[1][i][g], ok[o]), f"scenario {sc}: oracle differs"
    if (sc + 1) % 10 == 0:
        print(f"{sc + 1} scenarios ok, {time.time() - t0:.0f} s", flush=True)
print("SOAK OK", n_scen)
