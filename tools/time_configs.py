#!/usr/bin/env python3
"""Times the hot-path entry points on BASELINE.json's other configurations (C2, C3, C5) and the
stand-alone nearest kernel at C4.  Host-pointer API (PCIe-inclusive wall time) plus the per-kernel
device time from the library's HIP-event spans.  Prints one JSON line per case."""
import json
import math
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rrtqx_3d_amd import _capi, synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402


def spans(st):
    return {k: round(getattr(st, "ms_" + k), 4) for k in ("nn_scan", "nn_finish", "nn_nearest", "edges", "points", "dubins")}


def timed(ctx, fn, reps=5):
    fn()
    ctx.profile(2)
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    wall = (time.perf_counter() - t0) / reps
    st = ctx.stats()
    ctx.profile(0)
    return out, wall * 1e3, {k: round(v / reps, 4) for k, v in spans(st).items()}


def main():
    for name in ("C2", "C4", "C5s"):
        cfg = synth.CONFIGS[name]
        pts, Q, sph = synth.nodes(cfg.n_nodes, 3), synth.queries(cfg.batch, 3), synth.spheres(cfg.n_obstacles)
        r = synth.ball_radius(cfg.n_nodes, 3)
        with Context(3, node_capacity=cfg.n_nodes) as ctx:
            ctx.nodes_append(pts)
            ctx.spheres_set(sph)
            out, wall, k = timed(ctx, lambda: ctx.extend_candidates(Q, r, 0.5, cap=96 * cfg.batch))
            print(json.dumps({"case": f"{name} extend_candidates (host buffers)", "N": cfg.n_nodes, "M": cfg.n_obstacles,
                              "B": cfg.batch, "neighbors": int(len(out["idx"])), "wall_ms": round(wall, 3), "kernel_ms": k}))
            if name == "C4":
                _, wall, k = timed(ctx, lambda: ctx.nn_nearest(Q))
                print(json.dumps({"case": "C4 nn_nearest (full scan)", "wall_ms": round(wall, 3), "kernel_ms": k}))
                # the same candidate edges (both directions) against 256 random polygons, projected to (x, y)
                # as explicitEdgeCheck2D does (R/DRRT.jl:1536): the straight-edge polygon kernel at C4 scale
                owner = np.repeat(np.arange(cfg.batch), np.diff(out["offsets"]))
                p0 = np.concatenate([Q[owner], pts[out["idx"]]])
                p1 = np.concatenate([pts[out["idx"]], Q[owner]])
                ctx.polygons_set(synth.polygons(cfg.n_obstacles))
                (hit, _), wall, k = timed(ctx, lambda: ctx.edges_check(p0, p1, 0.5, kind=1))
                print(json.dumps({"case": "C4 edges_check vs 256 polygons (host buffers)", "edges": int(len(p0)),
                                  "hit_fraction": round(float(hit.mean()), 4), "wall_ms": round(wall, 3), "kernel_ms": k}))
                # and the fused preamble against the polygon list (edges formed on the device from the lists)
                ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
                out, wall, k = timed(ctx, lambda: ctx.extend_candidates(Q, r, 0.5, cap=96 * cfg.batch))
                ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 0)
                print(json.dumps({"case": "C4 extend_candidates vs 256 polygons (host buffers)", "neighbors": int(len(out["idx"])),
                                  "wall_ms": round(wall, 3), "kernel_ms": k}))
    cfg = synth.CONFIGS["C3"]
    pts, Q = synth.nodes(cfg.n_nodes, 4), synth.queries(cfg.batch, 4)
    polys = synth.polygons(cfg.n_obstacles)
    r = synth.ball_radius(cfg.n_nodes, 4, gamma=100.0, delta=10.0)
    with Context(4, node_capacity=cfg.n_nodes) as ctx:
        ctx.set_wrap(3, 2 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys)
        (off, idx, dist), wall, k = timed(ctx, lambda: ctx.nn_radius(Q, r, cap=4_000_000), reps=3)
        print(json.dumps({"case": "C3 nn_radius wrapped theta", "N": cfg.n_nodes, "B": cfg.batch, "r": r,
                          "neighbors": int(len(idx)), "wall_ms": round(wall, 3), "kernel_ms": k}))
        owner = np.repeat(np.arange(cfg.batch), np.diff(off))
        ne = min(len(idx), 262144)
        s, g = Q[owner[:ne]], pts[idx[:ne]]
        _, wall, k = timed(ctx, lambda: ctx.dubins_steer(s, g, 1.0), reps=3)
        print(json.dumps({"case": "C3 dubins_steer", "edges": ne, "wall_ms": round(wall, 3), "kernel_ms": k}))
        _, wall, k = timed(ctx, lambda: ctx.dubins_edges_check(s, g, 1.0, 0.5), reps=3)
        print(json.dumps({"case": "C3 dubins_edges_check (64 polygons)", "edges": ne, "wall_ms": round(wall, 3), "kernel_ms": k}))
        # the whole fused Dubins preamble at config scale: wrapped range search + both directed edges
        # of every neighbour steered and checked against the polygons
        out, wall, k = timed(ctx, lambda: ctx.extend_candidates_dubins(Q, r, 0.5, 1.0, cap=6_000_000), reps=2)
        print(json.dumps({"case": "C3 extend_candidates_dubins (host buffers)", "B": cfg.batch, "neighbors": int(len(out["idx"])),
                          "directed_edges": 2 * int(len(out["idx"])), "wall_ms": round(wall, 3), "kernel_ms": k}))


if __name__ == "__main__":
    main()
