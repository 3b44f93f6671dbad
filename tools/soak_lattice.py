"""Soak on a lattice: scenes whose every coordinate is a multiple of 1/4 -- axis-aligned boxes, right triangles
and diamonds, balls with lattice centres, edges between lattice points, nodes and samples on the lattice, search
radii that are distances between lattice points.  Random real scenes never put a point exactly on a line, a
distance exactly on a threshold or two segments on one line; here almost every test sits on such a boundary:
strict / non-strict comparisons, the "close to vertical" branches, segmentDistSqrd's coincident lines, the
inclusive root rule of the range search, ties of the nearest neighbour.  Everything through the C-ABI against
the oracle, bit for bit."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as O  # noqa: E402
from rrtqx_3d_amd import _capi  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402


def lattice(rng, span, shape):
    return rng.integers(-4 * span, 4 * span + 1, shape).astype(np.float64) / 4.0


def lattice_polygons(rng, m, span):
    polys, kinds = [], []
    for _ in range(m):
        c = lattice(rng, span, 2)
        w, h = rng.integers(1, 13, 2) / 4.0
        shape = int(rng.integers(0, 4))
        if shape == 0:
            v = [[0, 0], [w, 0], [w, h], [0, h]]                       # box, counter-clockwise
        elif shape == 1:
            v = [[0, 0], [0, h], [w, h], [w, 0]]                       # box, clockwise
        elif shape == 2:
            v = [[0, 0], [w, 0], [0, h]]                               # right triangle
        else:
            v = [[w, 0], [2 * w, h], [w, 2 * h], [0, h]]               # diamond (slopes +-h/w)
        polys.append(c + np.array(v, dtype=np.float64))
        kinds.append(1 if rng.uniform() < 0.15 else 3)
    return polys, kinds


def scene(sc):
    rng = np.random.default_rng(310_000 + sc)
    span = int(rng.choice([3, 6, 12]))
    out = {"edges": 0, "hits": 0, "nbrs": 0, "dubins_edges": 0, "dubins_flips": 0}
    # ---- polygons: edges, points ----
    m = int(rng.choice([1, 7, 33, 90]))
    polys, kinds = lattice_polygons(rng, m, span)
    ps = O.PolygonSet(polys, kinds=kinds)
    ne = 1200
    p0 = np.zeros((ne, 3)); p1 = np.zeros((ne, 3))
    p0[:, :2] = lattice(rng, span, (ne, 2))
    step = rng.integers(-12, 13, (ne, 2)).astype(np.float64) / 4.0
    step[: ne // 3, rng.integers(0, 2)] = 0.0                           # a third axis-aligned
    p1[:, :2] = p0[:, :2] + step
    rr = float(rng.choice([0.0, 0.25, 0.5, 1.0]))
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.polygons_set(polys, kinds=kinds)
        hit, first = ctx.edges_check(p0, p1, rr, kind=1)
        oh, of = O.edges_check_polygons(ps, p0, p1, rr)
        assert np.array_equal(hit, oh) and np.array_equal(first, of), f"scene {sc}: polygon edges differ"
        pts = p0[:300]
        unsafe, clr = ctx.points_check(pts, rr, kind=1)
        exp = [O.point_check_polygons(ps, p, rr) for p in pts]
        assert np.array_equal(unsafe.astype(bool), np.array([e[0] for e in exp])), f"scene {sc}: polygon points differ"
        assert np.array_equal(clr, np.array([e[1] for e in exp])), f"scene {sc}: polygon clearances differ"
        flag_only, _ = ctx.points_check(pts, rr, kind=1, want_clearance=False)
        assert np.array_equal(flag_only, unsafe), f"scene {sc}: flag-only polygon point check differs"
        out["edges"] += ne; out["hits"] += int(hit.sum())
    # ---- the same list with obstacles that move in time (kinds 6 / 7): path rows and times on the lattice, edges
    # and points whose third coordinate (time) falls exactly on path times ----
    mk = list(kinds)
    paths = [None] * m
    for jj in range(m):
        if rng.uniform() < 0.3:
            mk[jj] = int(rng.choice([6, 7]))
            rows = int(rng.integers(1, 6))
            paths[jj] = np.c_[lattice(rng, 2, (rows, 2)), np.sort(rng.integers(0, 9, rows)).astype(np.float64)]
    act_m = [int(rng.uniform() > 0.1) for _ in range(m)]
    psm = O.PolygonSet(polys, kinds=mk, active=act_m, paths=paths)
    p0[:, 2] = rng.integers(-1, 10, ne); p1[:, 2] = p0[:, 2] + rng.integers(-3, 4, ne)
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.polygons_set(polys, kinds=mk, active=act_m, paths=paths)
        hit, first = ctx.edges_check(p0, p1, rr, kind=1)
        oh, of = O.edges_check_polygons(psm, p0, p1, rr)
        assert np.array_equal(hit, oh) and np.array_equal(first, of), f"scene {sc}: edges against moving obstacles differ"
        pts = p0[:300]
        unsafe, clr = ctx.points_check(pts, rr, kind=1)
        exp = [O.point_check_polygons(psm, p, rr) for p in pts]
        assert np.array_equal(unsafe.astype(bool), np.array([e[0] for e in exp])), f"scene {sc}: points against moving obstacles differ"
        assert np.array_equal(clr, np.array([e[1] for e in exp])), f"scene {sc}: clearances against moving obstacles differ"
        out["edges"] += ne; out["hits"] += int(hit.sum())
    # ---- Dubins edges between lattice poses (headings multiples of pi / 4).  Counted, not asserted: the arcs come
    # out of sin / cos / atan2, which differ in the last bit between the host's libm, the device's and Julia's, and
    # on a lattice that bit decides things -- a piece that grazes a side (boolean), or an arc of length exactly 0
    # that comes out of the mod 2 pi as a full turn when the difference rounds to -1e-16 (a goal exactly ahead on
    # the start's heading line: the cost then differs by 2 pi r_min).  Neither side is "right" without the
    # reference's own libm; the counts are bounded in the test so that a real defect would show ----
    nd = 300
    ds = np.zeros((nd, 4)); dg = np.zeros((nd, 4))
    ds[:, :2] = lattice(rng, span, (nd, 2)); ds[:, 3] = rng.integers(0, 8, nd) * (np.pi / 4)
    dg[:, :2] = ds[:, :2] + rng.integers(-16, 17, (nd, 2)) / 4.0; dg[:, 3] = rng.integers(0, 8, nd) * (np.pi / 4)
    r_min = float(rng.choice([0.5, 1.0, 2.0]))
    with Context(4) as ctx:
        ctx.nodes_append(ds[:2])
        ctx.polygons_set(polys, kinds=kinds)
        cost, word, dhit, tl = ctx.dubins_edges_check(ds, dg, r_min, rr)
        toff, txy = ctx.dubins_trajectory(ds, dg, r_min)
    # the check itself is exact: fed the DEVICE's polyline, the oracle's two-stage check gives the device's boolean
    for i2 in range(nd):
        if np.isfinite(cost[i2]) and toff[i2 + 1] > toff[i2]:
            h_own, _ = O.dubins_edge_check_polygons(ps, ds[i2], dg[i2], txy[toff[i2]:toff[i2 + 1]], rr, r_min)
            if bool(dhit[i2]) != bool(h_own):
                print(f"scene {sc}: Dubins check differs on the device's own polyline: s={ds[i2].tolist()} g={dg[i2].tolist()} "
                      f"r_min={r_min} rr={rr} device {dhit[i2]} oracle {h_own}", flush=True)
                out["dubins_own_flips"] = out.get("dubins_own_flips", 0) + 1
    flips = 0
    for i2 in range(nd):
        c_o, w_o, traj = O.dubins_steer(ds[i2], dg[i2], r_min)
        h_o, _ = O.dubins_edge_check_polygons(ps, ds[i2], dg[i2], traj, rr, r_min)
        if not (cost[i2] == c_o or (np.isnan(cost[i2]) and np.isnan(c_o))) or bytes(word[i2]).decode() != w_o or tl[i2] != len(traj):
            print(f"scene {sc}: Dubins cost differs: s={ds[i2].tolist()} g={dg[i2].tolist()} r_min={r_min} "
                  f"device {cost[i2]!r} {bytes(word[i2]).decode() if hasattr(word[i2], '__len__') else word[i2]} oracle {c_o!r} {w_o}", flush=True)
            out["dubins_cost_diffs"] = out.get("dubins_cost_diffs", 0) + 1
        flips += int(bool(dhit[i2]) != bool(h_o))
    out["dubins_edges"] = nd; out["dubins_flips"] = flips
    # ---- balls in 3-D: edges, points ----
    ms = int(rng.choice([1, 30, 200]))
    sph = np.c_[lattice(rng, span, (ms, 3)), rng.integers(1, 9, ms) / 4.0]
    act = (rng.uniform(size=ms) > 0.1).astype(np.uint8)
    osph, mo = O.make_spheres(sph, act)
    q0 = lattice(rng, span, (ne, 3))
    q1 = q0 + rng.integers(-8, 9, (ne, 3)) / 4.0
    q1[:15] = q0[:15]                                                   # zero-length edges
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.spheres_set(sph, act)
        hit, first = ctx.edges_check(q0, q1, rr)
        oh, of = O.edges_check_spheres(osph, mo, q0, q1, rr)
        assert np.array_equal(hit, oh) and np.array_equal(first, of), f"scene {sc}: sphere edges differ"
        for quick in (True, False):
            unsafe, clr = ctx.points_check(q0[:400], rr, quick=quick)
            ou, oc = O.points_check_spheres(osph, mo, q0[:400], rr, quick=quick)
            assert np.array_equal(unsafe, ou) and np.array_equal(clr, oc), f"scene {sc}: sphere points differ"
        out["edges"] += ne; out["hits"] += int(hit.sum())
    # ---- search on the lattice: distances exactly on the radius, ties of the nearest neighbour ----
    n = int(rng.choice([300, 3000, 20000]))
    nodes = np.unique(lattice(rng, span, (n, 3)), axis=0)
    nodes = nodes[rng.permutation(len(nodes))]
    tree = O.KDTree(3)
    tree.insert_many(nodes)
    Q = lattice(rng, span, (64, 3))
    r = float(rng.choice([0.75, 1.25, 2.5, 3.25]))                      # 3-4-5 and 5-12-13 triples in quarters
    with Context(3) as ctx:
        ctx.spheres_set(sph, act)
        ctx.nodes_append(nodes)
        offsets, idx, dist = ctx.nn_radius(Q, r)
        for i, q in enumerate(Q):
            oi, ok = tree.within_range(r, q)
            o = np.argsort(oi, kind="stable")
            a, b = offsets[i], offsets[i + 1]
            assert np.array_equal(idx[a:b], oi[o]) and np.array_equal(dist[a:b], ok[o]), f"scene {sc}: range search differs"
        out["nbrs"] += len(idx)
        ex = ctx.extend_candidates(Q, r, rr)
        assert np.array_equal(ex["offsets"], offsets) and np.array_equal(ex["idx"], idx), f"scene {sc}: fused lists differ"
        # both directed edges of every neighbour, as the reference checks them one by one
        nbr = nodes[idx]
        own = np.repeat(np.arange(len(Q)), np.diff(offsets))
        ho, _ = O.edges_check_spheres(osph, mo, Q[own], nbr, rr)
        hi, _ = O.edges_check_spheres(osph, mo, nbr, Q[own], rr)
        assert np.array_equal(ex["hit_out"], ho) and np.array_equal(ex["hit_in"], hi), f"scene {sc}: fused edge flags differ"
        su, _ = O.points_check_spheres(osph, mo, Q, rr, quick=True)
        assert np.array_equal(ex["sample_unsafe"], su), f"scene {sc}: fused sample flags differ"
        nidx, ndist = ctx.nn_nearest(Q)
        for i, q in enumerate(Q):
            ri, rd = tree.nearest(q)
            assert ndist[i] == rd, f"scene {sc}: nearest distance differs"
            # ties: the reference keeps the first it meets in its tree walk, the device the lowest index (documented)
            d_all = np.sqrt(((nodes - q) ** 2).sum(axis=1))
            assert d_all[nidx[i]] == d_all.min()
    # ---- the fused preamble against the polygon list (RRTX_OPT_EXTEND_OBSTACLES = 1): its edges come straight
    # from the lists and walk only the obstacles near their samples; held to the oracle edge by edge ----
    with Context(3) as ctx:
        ctx.nodes_append(nodes)
        ctx.polygons_set(polys, kinds=mk, active=act_m, paths=paths)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        Qt = Q.copy(); Qt[:, 2] = rng.integers(0, 9, len(Q))
        ex = ctx.extend_candidates(Qt, r, rr)
        own = np.repeat(np.arange(len(Qt)), np.diff(ex["offsets"]))
        nbr = nodes[ex["idx"]]
        ho, _ = O.edges_check_polygons(psm, Qt[own], nbr, rr)
        hi, _ = O.edges_check_polygons(psm, nbr, Qt[own], rr)
        assert np.array_equal(ex["hit_out"], ho) and np.array_equal(ex["hit_in"], hi), f"scene {sc}: fused polygon edge flags differ"
        exp = np.array([O.point_check_polygons(psm, q, rr)[0] for q in Qt])
        assert np.array_equal(ex["sample_unsafe"].astype(bool), exp), f"scene {sc}: fused polygon sample flags differ"
    # ---- the same in the Dubins space [x y 0 theta], theta wrapped at 2 pi (R/DRRT.jl:3312): headings multiples
    # of pi / 4 (0 and 2 pi - the wrap point itself - included), ghost copies, distances exactly on the radius ----
    n4 = int(rng.choice([200, 2500]))
    nodes4 = np.zeros((n4, 4))
    nodes4[:, :2] = lattice(rng, span, (n4, 2))
    nodes4[:, 3] = rng.integers(0, 9, n4) * (np.pi / 4)
    nodes4 = np.unique(nodes4, axis=0)
    nodes4 = nodes4[rng.permutation(len(nodes4))]
    tree4 = O.KDTree(4, wraps=[3], wrap_points=[2.0 * np.pi])
    tree4.insert_many(nodes4)
    Q4 = np.zeros((48, 4))
    Q4[:, :2] = lattice(rng, span, (48, 2))
    Q4[:, 3] = rng.integers(0, 9, 48) * (np.pi / 4)
    r4 = float(rng.choice([0.75, 1.25, np.pi / 4, np.pi / 2, 3.25]))
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * np.pi)
        ctx.nodes_append(nodes4)
        offsets, idx, dist = ctx.nn_radius(Q4, r4)
        for i, q in enumerate(Q4):
            oi, ok = tree4.within_range(r4, q)
            o = np.argsort(oi, kind="stable")
            a, b = offsets[i], offsets[i + 1]
            assert np.array_equal(idx[a:b], oi[o]) and np.array_equal(dist[a:b], ok[o]), f"scene {sc}: wrapped range search differs"
        out["nbrs"] += len(idx)
        nidx, ndist = ctx.nn_nearest(Q4)
        for i, q in enumerate(Q4):
            ri, rd = tree4.nearest(q)
            assert ndist[i] == rd, f"scene {sc}: wrapped nearest distance differs"
        # the fused Dubins preamble against the stand-alone entry points (same device arithmetic: bit for bit)
        ctx.polygons_set(polys, kinds=kinds)
        exd = ctx.extend_candidates_dubins(Q4, r4, rr, r_min)
        assert np.array_equal(exd["offsets"], offsets) and np.array_equal(exd["idx"], idx), f"scene {sc}: fused Dubins lists differ"
        own = np.repeat(np.arange(len(Q4)), np.diff(offsets))
        if len(idx):
            c_o, w_o, h_o, _ = ctx.dubins_edges_check(Q4[own], nodes4[idx], r_min, rr)
            c_i, w_i, h_i, _ = ctx.dubins_edges_check(nodes4[idx], Q4[own], r_min, rr)
            assert np.array_equal(exd["cost_out"], c_o) and np.array_equal(exd["cost_in"], c_i), f"scene {sc}: fused Dubins costs differ"
            assert np.array_equal(exd["hit_out"], h_o) and np.array_equal(exd["hit_in"], h_i), f"scene {sc}: fused Dubins flags differ"
    return out


if __name__ == "__main__":
    n_scen = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    t0 = time.time()
    tot = {"edges": 0, "hits": 0, "nbrs": 0, "dubins_edges": 0, "dubins_flips": 0}
    for sc in range(n_scen):
        o = scene(sc)
        for k in o:
            tot[k] = tot.get(k, 0) + o[k]
        if (sc + 1) % 10 == 0:
            print(f"{sc + 1} scenes ok, {tot}, {time.time() - t0:.0f} s", flush=True)
    print("SOAK OK", n_scen, tot)
