"""Dubins edges in a space WITH time ([x y t theta], CSpace.spaceHasTime): edge.dist / Wdist / velocity,
validMove, the time-stamped trajectory (R/DRRT_DubinsEdge_functions.jl:115-121, 660-697) and the
two-stage edge check against static polygons and polygons that move in time (kinds 6 / 7,
:750-774 + R/DRRT.jl:1579-1651), through the C-ABI against the oracle.  Costs and times within the
exact since round 3 (shared deterministic transcendentals, include/rrtx_detmath.h); was: words / booleans may differ only on numerical
ties (the tests bound the count).  Moving obstacles: the reference's own
environments/rand_StaticTime_7.txt (tests/golden/env_inputs.json)."""
import json
import math
import os

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, synth
from rrtqx_3d_amd._capi import RrtxError
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.abspath(__file__))
RR, RMIN = 0.5, 2.0            # R/dubinsExperimentsForPaper.jl:225 (minTurningRadius = 2.0)
VMIN, VMAX = 5.0, 30.0         # :222-223


def _env():
    d = json.load(open(os.path.join(ROOT, "golden", "env_inputs.json")))
    polys = [np.array(p, dtype=np.float64) for p in d["rand_StaticTime_7_polygons"]]
    paths = [np.array(p, dtype=np.float64) for p in d["rand_StaticTime_7_paths"]]
    return polys[::-1], paths[::-1]


def _edges(rng, n, spread=14.0):
    s = np.zeros((n, 4)); g = np.zeros((n, 4))
    s[:, :2] = rng.uniform(-45, 45, (n, 2))
    g[:, :2] = s[:, :2] + rng.normal(0, spread, (n, 2))
    s[:, 2] = rng.uniform(12, 35, n)
    g[:, 2] = s[:, 2] - rng.uniform(0.05, 4.0, n)            # planning runs in reverse time
    g[: n // 10, 2] = s[: n // 10, 2] + rng.uniform(0.0, 2.0, n // 10)   # some edges the wrong way (validMove false)
    s[:, 3] = rng.uniform(0, 2 * math.pi, n); g[:, 3] = rng.uniform(0, 2 * math.pi, n)
    return s, g


def test_steer_full_and_trajectory_with_time(oracle):
    rng = np.random.default_rng(7)
    s, g = _edges(rng, 3000)
    with Context(4) as ctx:
        ctx.set_space_has_time(True)
        ctx.set_dubins_velocity(VMIN, VMAX)
        out = ctx.dubins_steer_full(s, g, RMIN)
        cost, _ = ctx.dubins_steer(s, g, RMIN)
        assert np.array_equal(cost, out["dist"])                     # rrtx_dubins_steer returns edge.dist
        off, traj = ctx.dubins_trajectory(s[:400], g[:400], RMIN)
        assert traj.shape[1] == 3
        for k in range(len(s)):
            d, w, v, word, tr = oracle.dubins_steer_time(s[k], g[k], RMIN, piecewise=True)
            assert out["wdist"][k] == w and out["dist"][k] == d and out["velocity"][k] == v, k
            assert out["word"][k].decode() == word, k
            assert bool(out["valid_move"][k]) == oracle.dubins_valid_move_time(s[k], g[k], v, VMIN, VMAX), k
            if k < 400:
                mine = traj[off[k]:off[k + 1]]
                assert np.array_equal(mine, tr), k          # (x, y, t) rows bit for bit
                assert mine[0, 2] == s[k, 2] and np.array_equal(mine[-1], g[k, :3])
                # the reference's own running sum (oracle default) differs from the piecewise time column by rounding only
                tr_ref = oracle.dubins_steer_time(s[k], g[k], RMIN)[4]
                assert np.array_equal(tr_ref[:, :2], tr[:, :2]) and np.allclose(tr_ref[:, 2], tr[:, 2], rtol=1e-12, atol=1e-12)
        assert 0.2 < out["valid_move"].mean() < 0.95
        # without the option the same calls are the space-without-time ones
        ctx.set_space_has_time(False)
        c0, _ = ctx.dubins_steer(s, g, RMIN)
        assert np.array_equal(c0, out["wdist"])
        off2, traj2 = ctx.dubins_trajectory(s[:50], g[:50], RMIN)
        assert traj2.shape[1] == 2 and np.array_equal(off2, off[:51])


def test_dubins_edges_check_with_time_static_and_moving(oracle):
    polys, paths = _env()
    m = len(polys)
    rng = np.random.default_rng(19)
    stat = synth.polygons(24, seed=5)
    all_polys = polys + stat
    kinds = [6 if i % 2 else 7 for i in range(m)] + [3] * len(stat)
    all_paths = paths + [None] * len(stat)
    active = np.ones(len(all_polys), dtype=np.uint8)
    active[[2, m + 3]] = 0
    ps = oracle.PolygonSet(all_polys, kinds=kinds, paths=all_paths, active=active)
    s, g = _edges(rng, 2500, spread=20.0)
    t_hi = max(p[:, 2].max() for p in paths)
    s[:, 2] = rng.uniform(0.0, t_hi, len(s)); g[:, 2] = s[:, 2] - rng.uniform(0.05, 6.0, len(s))
    with Context(4) as ctx:
        ctx.polygons_set(all_polys, kinds=kinds, paths=all_paths, active=active)
        with pytest.raises(RrtxError):                       # pieces without time stamps cannot meet a moving obstacle
            ctx.dubins_edges_check(s[:8], g[:8], RMIN, RR)
        ctx.set_space_has_time(True)
        cost, word, hit, tl = ctx.dubins_edges_check(s, g, RMIN, RR)
        ref_flips = 0
        for k in range(len(s)):
            d, w, v, wd, tr = oracle.dubins_steer_time(s[k], g[k], RMIN, piecewise=True)
            assert cost[k] == d and word[k].decode() == wd and tl[k] == len(tr), k
            h, _ = oracle.dubins_edge_check_polygons_time(ps, s[k], g[k], tr, RR, RMIN)
            assert bool(hit[k]) == h, k
            if k % 5 == 0:      # the reference's running-sum time column gives the same booleans on these scenes
                tr_ref = oracle.dubins_steer_time(s[k], g[k], RMIN)[4]
                ref_flips += oracle.dubins_edge_check_polygons_time(ps, s[k], g[k], tr_ref, RR, RMIN)[0] != h
        assert ref_flips == 0
        assert 0.05 < hit.mean() < 0.9
        # the moving ones matter: with them switched off fewer edges collide
        act2 = active.copy(); act2[:m] = 0
        ctx.polygons_set(all_polys, kinds=kinds, paths=all_paths, active=act2)
        _, _, hit2, _ = ctx.dubins_edges_check(s, g, RMIN, RR)
        assert hit2.sum() < hit.sum() and not (hit2 & ~hit).any()


def test_fused_dubins_preamble_with_time(oracle):
    polys, paths = _env()
    m = len(polys)
    kinds = [6] * m
    ps = oracle.PolygonSet(polys, kinds=kinds, paths=paths)
    rng = np.random.default_rng(23)
    n, nq, r = 6000, 40, 9.0
    pts = synth.nodes(n, 4)
    pts[:, 2] = rng.uniform(10.0, 35.0, n)                   # the time coordinate (R/dubinsExperimentsForPaper.jl:209-210)
    Q = synth.queries(nq, 4)
    Q[:, 2] = rng.uniform(10.0, 35.0, nq)
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, kinds=kinds, paths=paths)
        ctx.set_space_has_time(True)
        ctx.set_dubins_velocity(VMIN, VMAX)
        out = ctx.extend_candidates_dubins(Q, r, RR, RMIN)
        off, idx = out["offsets"], out["idx"]
        assert len(idx) > 200
        for i in range(nq):
            ri, rk = tree.within_range(r, Q[i])
            o = np.argsort(ri)
            assert np.array_equal(idx[off[i]:off[i + 1]], ri[o]) and np.array_equal(out["key"][off[i]:off[i + 1]], rk[o])
            for e in range(off[i], off[i + 1]):
                for (a, b, ck, hk) in ((Q[i], pts[idx[e]], "cost_out", "hit_out"), (pts[idx[e]], Q[i], "cost_in", "hit_in")):
                    d, w, v, wd, tr = oracle.dubins_steer_time(a, b, RMIN, piecewise=True)
                    assert out[ck][e] == d, (e, ck)
                    h, _ = oracle.dubins_edge_check_polygons_time(ps, a, b, tr, RR, RMIN)
                    bad = not oracle.dubins_valid_move_time(a, b, v, VMIN, VMAX)
                    assert int(out[hk][e]) == (int(h) | (2 if bad else 0)), (e, hk)
        both = out["hit_out"].astype(int) | out["hit_in"].astype(int)
        assert (both & 2).any() and (out["hit_out"] & 1).any()   # invalid moves and collisions both occur


def test_trajectory_row_width_is_the_contexts():
    """rrtx_dubins_trajectory writes 2 or 3 doubles per row by the CONTEXT's RRTX_OPT_SPACE_HAS_TIME; the
    caller states the width its buffer was sized for and a mismatch is an error, not an overrun.  The Python wrapper
    reads the width from the context (rrtx_get_option), so setting the option directly cannot desynchronise it."""
    import ctypes as C
    rng = np.random.default_rng(3)
    s, g = _edges(rng, 64)
    with Context(4) as ctx:
        ctx.set_option(_capi.RRTX_OPT_SPACE_HAS_TIME, 1)            # not through set_space_has_time
        assert ctx.space_has_time and ctx.get_option(_capi.RRTX_OPT_SPACE_HAS_TIME) == 1
        off, traj = ctx.dubins_trajectory(s, g, RMIN)
        assert traj.shape[1] == 3 and off[-1] == len(traj) and np.array_equal(traj[0, :3], s[0, :3])
        # a caller that still believes in two columns is refused and its buffer left alone
        offs = np.zeros(len(s) + 1, dtype=np.int64)
        buf = np.full((int(off[-1]), 2), -7.0)
        needed = C.c_int64()
        rc = ctx._lib.rrtx_dubins_trajectory(ctx._h, _capi._ptr(s), _capi._ptr(g), len(s), RMIN, _capi._ptr(offs),
                                             _capi._ptr(buf), 2, len(buf), C.byref(needed))
        assert rc == _capi.RRTX_E_INVALID and (buf == -7.0).all()
        ctx.set_option(_capi.RRTX_OPT_SPACE_HAS_TIME, 0)
        off2, traj2 = ctx.dubins_trajectory(s, g, RMIN)
        assert traj2.shape[1] == 2 and np.array_equal(off2, off)
        with pytest.raises(RrtxError):
            ctx.get_option(999)
