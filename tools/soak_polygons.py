"""One-off soak: random polygon scenes (static / ball / moving / inactive obstacles, 1..700 of them),
edges and points through the C-ABI against the oracle: hit flags, first-hit indices, unsafe flags and
clearances must be identical."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 40
t0 = time.time()
n_edges = n_hits = 0
for sc in range(n_scen):
    rng = np.random.default_rng(90_000 + sc)
    m = int(rng.choice([1, 5, 31, 32, 33, 64, 100, 257, 700]))
    span = float(rng.choice([10.0, 50.0, 1000.0]))
    size = span * float(rng.choice([0.02, 0.1, 0.4]))
    polys, kinds, paths, active = [], [], [], []
    for i in range(m):
        c = rng.uniform(-span, span, 2)
        nv = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
        polys.append(c + np.c_[np.cos(ang), np.sin(ang)] * rng.uniform(0.2, 1.0, (nv, 1)) * size)
        k = int(rng.choice([3, 3, 3, 1, 6, 7]))
        kinds.append(k)
        rows = int(rng.integers(1, 7))
        paths.append(np.c_[rng.uniform(-size, size, (rows, 2)), np.sort(rng.uniform(0, 30, rows))] if k in (6, 7) else None)
        active.append(int(rng.uniform() > 0.15))
    ps = O.PolygonSet(polys, kinds=kinds, active=active, paths=paths)
    ne = 1500
    dim = 3 if rng.uniform() < 0.7 else 4
    p0 = np.zeros((ne, dim)); p1 = np.zeros((ne, dim))
    p0[:, :2] = rng.uniform(-span, span, (ne, 2))
    p1[:, :2] = p0[:, :2] + rng.normal(0, size * float(rng.choice([0.1, 1.0, 5.0])), (ne, 2))
    p0[:, 2] = rng.uniform(-3, 33, ne); p1[:, 2] = p0[:, 2] + rng.normal(0, 5, ne)
    p1[:20] = p0[:20]
    rr = float(rng.choice([0.0, 0.05, 0.5, 2.0])) * size / 5 + (0.0 if rng.uniform() < 0.8 else -0.1)
    with Context(dim) as ctx:
        ctx.polygons_set(polys, kinds=kinds, active=active, paths=paths)
        hit, first = ctx.edges_check(p0, p1, rr, kind=1)
        oh, of = O.edges_check_polygons(ps, p0, p1, rr)
        assert np.array_equal(hit, oh) and np.array_equal(first, of), f"scenario {sc}: edges differ"
        pts = p0[:400]
        unsafe, clr = ctx.points_check(pts, abs(rr), kind=1)
        exp = [O.point_check_polygons(ps, p, abs(rr)) for p in pts]
        assert np.array_equal(unsafe.astype(bool), np.array([e[0] for e in exp])), f"scenario {sc}: unsafe differs"
        assert np.array_equal(clr, np.array([e[1] for e in exp])), f"scenario {sc}: clearance differs"
    n_edges += ne; n_hits += int(hit.sum())
    if (sc + 1) % 10 == 0:
        print(f"{sc + 1} scenarios ok, {n_edges} edges, {n_hits} hits, {time.time() - t0:.0f} s", flush=True)
print("SOAK OK", n_scen, n_edges, n_hits)
