"""One-off soak: Dubins edge checks (two-stage, dealt across the wave) against the oracle on random
scenes: costs, words, row counts and collision booleans equal bit for bit (shared include/rrtx_detmath.h)."""
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.time()
tot = flips = hits = 0
for sc in range(n_scen):
    rng = np.random.default_rng(30_000 + sc)
    m = int(rng.choice([1, 7, 63, 64, 65, 130]))
    span = float(rng.choice([20.0, 50.0]))
    polys = []
    for _ in range(m):
        c = rng.uniform(-span, span, 2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, int(rng.integers(3, 7))))
        polys.append(c + np.c_[np.cos(ang), np.sin(ang)] * rng.uniform(0.5, span / 5))
    kinds = [1 if rng.uniform() < 0.1 else 3 for _ in range(m)]
    active = [int(rng.uniform() > 0.1) for _ in range(m)]
    ps = O.PolygonSet(polys, kinds=kinds, active=active)
    ne = 1200
    s = np.zeros((ne, 4)); g = np.zeros((ne, 4))
    s[:, :2] = rng.uniform(-span, span, (ne, 2)); s[:, 3] = rng.uniform(0, 2 * math.pi, ne)
    g[:, :2] = s[:, :2] + rng.normal(0, float(rng.choice([1.0, 5.0, 15.0])), (ne, 2)); g[:, 3] = rng.uniform(0, 2 * math.pi, ne)
    r_min = float(rng.choice([0.5, 1.0, 2.0]))
    rr = float(rng.choice([0.0, 0.3, 1.0]))
    with Context(4) as ctx:
        ctx.nodes_append(s[:2])
        ctx.polygons_set(polys, kinds=kinds, active=active)
        cost, word, hit, tl = ctx.dubins_edges_check(s, g, r_min, rr)
    for i in range(ne):
        c, w, traj = O.dubins_steer(s[i], g[i], r_min)
        h, _ = O.dubins_edge_check_polygons(ps, s[i], g[i], traj, rr, r_min)
        assert cost[i] == c and bytes(word[i]).decode() == w and tl[i] == len(traj), (sc, i, cost[i], c, word[i], w)
        flips += (bool(hit[i]) != h)
    tot += ne; hits += int(hit.sum())
    if (sc + 1) % 5 == 0:
        print(f"{sc + 1} scenarios, {tot} edges, {hits} hits, {flips} boolean flips, {time.time() - t0:.0f} s", flush=True)
assert flips == 0, flips
print("SOAK OK", n_scen, tot, hits, flips)
