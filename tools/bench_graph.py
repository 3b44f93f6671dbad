"""Time the device cost propagation (rrtx_graph_cost_to_root) on a C4-shaped planner graph and, on a bounded
sample, the oracle's reduceInconsistency beside it.  Run on the GPU box:

    python tools/bench_graph.py [--nodes 200000] [--out gpurun_out/graph_cost.json]

The graph is the one extend() builds: both directed edges between every pair of nodes closer than the shrinking
ball radius (found with the device range search), costs = SimpleEdge distances."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401  (first: one HIP runtime image in the process)

from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402


def build_edges(ctx, pts, r, chunk=65536):
    s, e = [], []
    for a in range(0, len(pts), chunk):
        off, idx, _ = ctx.nn_radius(pts[a:a + chunk], r)
        own = np.repeat(np.arange(a, a + len(off) - 1), np.diff(off))
        keep = own != idx
        s.append(own[keep].astype(np.int32))
        e.append(idx[keep])
    return np.concatenate(s), np.concatenate(e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=200_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-nodes", type=int, default=20_000)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    n = a.nodes
    batch = 16384
    pts_all = synth.nodes(n + batch, 3)
    pts = pts_all[:n]
    r = synth.ball_radius(n, 3)
    res = {"n_nodes": n, "ball_radius": r}
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        s, e = build_edges(ctx, pts, r)
        ctx.graph_edges_append(s, e)
        res["n_edges"] = int(len(s))
        lmc, par, passes = ctx.graph_cost_to_root(0)
        ts = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            lmc, par, passes = ctx.graph_cost_to_root(0)
            ts.append(time.perf_counter() - t0)
        res.update(passes=passes, ms_full_solve=1e3 * min(ts), reachable=int(np.isfinite(lmc).sum()),
                   max_cost=float(np.nanmax(lmc[np.isfinite(lmc)])))
        t0 = time.perf_counter()
        ctx.graph_cost_update(0)
        res["ms_update_nothing_new"] = 1e3 * (time.perf_counter() - t0)
        # an obstacle appears: sweep, block, update
        sph = np.array([[pts[0, 0] + 6.0, pts[0, 1], pts[0, 2], 5.0]])
        ctx.spheres_set(sph, np.ones(1, dtype=np.uint8))
        t0 = time.perf_counter()
        ids = ctx.obstacle_sweep(0, 0.5 + r + 5.0, 0.5)
        t1 = time.perf_counter()
        ctx.graph_edges_block(ids)
        t2 = time.perf_counter()
        lmc2, _, p2 = ctx.graph_cost_update(0)
        t3 = time.perf_counter()
        res.update(ms_sweep=1e3 * (t1 - t0), ms_block=1e3 * (t2 - t1), ms_update_after_block=1e3 * (t3 - t2),
                   blocked=int(len(ids)), passes_after_block=p2, nodes_changed=int((lmc2 != lmc).sum()))
        lmc_full, _, _ = ctx.graph_cost_to_root(0)
        res["update_equals_full_after_block"] = bool(np.array_equal(lmc_full, lmc2))
        # the tree grows by one batch of extends: new nodes, both directed edges to their neighbours
        new = pts_all[n:]
        off, idx, _ = ctx.nn_radius(new, r)
        own = np.repeat(np.arange(n, n + batch), np.diff(off)).astype(np.int32)
        ctx.nodes_append(new)
        t0 = time.perf_counter()
        ctx.graph_edges_append(np.concatenate([own, idx]), np.concatenate([idx, own]))
        t1 = time.perf_counter()
        lmc3, _, p3 = ctx.graph_cost_update(0)
        t2 = time.perf_counter()
        res.update(batch=batch, batch_edges=int(2 * len(own)), ms_append_edges=1e3 * (t1 - t0), ms_update_after_batch=1e3 * (t2 - t1),
                   passes_after_batch=p3, nodes_improved_by_batch=int((lmc3[:n] < lmc2).sum()))
        lmc_full, _, _ = ctx.graph_cost_to_root(0)
        res["update_equals_full_after_batch"] = bool(np.array_equal(lmc_full, lmc3))
    # the oracle on a smaller graph of the same density (bounded CPU sample)
    from oracle import oracle as orc
    m = a.cpu_nodes
    sub = pts[:m] * (m / n) ** (1.0 / 3.0)               # same density, smaller box
    with Context(3) as ctx:
        ctx.nodes_append(sub)
        s2, e2 = build_edges(ctx, sub, r)
        ctx.graph_edges_append(s2, e2)
        t0 = time.perf_counter()
        got, _, _ = ctx.graph_cost_to_root(0)
        t_dev = time.perf_counter() - t0
    d = sub[s2] - sub[e2]
    w = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2])
    g = orc.Graph(m + 1)
    for x, y, c in zip(s2.tolist(), e2.tolist(), w.tolist()):
        g.add_edge(x, y, c)
    for v in range(m + 1):
        g.set_node(v, float("inf"), float("inf"))
    g.set_node(0, 0.0, float("inf"))
    g.verifyInQueue(0)
    t0 = time.perf_counter()
    g.reduceInconsistency(m, 0)
    t_cpu = time.perf_counter() - t0
    res["cpu_sample"] = {"n_nodes": m, "n_edges": int(len(s2)), "oracle_ms": 1e3 * t_cpu, "device_ms": 1e3 * t_dev,
                         "identical": bool(np.array_equal(g.lmc()[:m], got))}
    line = json.dumps(res)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")


if __name__ == "__main__":
    main()
