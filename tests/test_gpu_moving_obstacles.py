"""Polygons that move in time (Obstacle kinds 6 and 7): explicitEdgeCheck2D's closest-approach test
(R/DRRT_Q.jl:1699-1771) and the time-shifted point checks (R/DRRT.jl:1289-1305, 1395-1420) through the
C-ABI against the oracle, bit-exact booleans / first-hit indices / clearances.  Inputs: the
reference's own environments/rand_StaticTime_7.txt (13 moving polygons, committed as numbers in
tests/golden/env_inputs.json) and seeded random scenes."""
import json
import os

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, drrt, envio, synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.abspath(__file__))


def _env():
    d = json.load(open(os.path.join(ROOT, "golden", "env_inputs.json")))
    polys = [np.array(p, dtype=np.float64) for p in d["rand_StaticTime_7_polygons"]]
    paths = [np.array(p, dtype=np.float64) for p in d["rand_StaticTime_7_paths"]]
    return polys[::-1], paths[::-1]                  # list order = reverse file order (listPush)


def _edges(rng, n, lo, hi, t_hi, dim):
    p0 = np.zeros((n, dim)); p1 = np.zeros((n, dim))
    p0[:, :2] = rng.uniform(lo, hi, (n, 2))
    p1[:, :2] = p0[:, :2] + rng.normal(0, 12.0, (n, 2))
    p0[:, 2] = rng.uniform(-5, t_hi + 5, n)
    p1[:, 2] = p0[:, 2] + rng.normal(0, 6.0, n)      # either direction in time
    if dim == 4:
        p0[:, 3] = rng.uniform(0, 2 * np.pi, n); p1[:, 3] = rng.uniform(0, 2 * np.pi, n)
    return p0, p1


@pytest.mark.parametrize("dim", [3, 4])
def test_reference_env_moving_polygons(oracle, dim):
    polys, paths = _env()
    m = len(polys)
    kinds = [6 if i % 3 else 7 for i in range(m)]
    ps = oracle.PolygonSet(polys, kinds=kinds, paths=paths)
    rng = np.random.default_rng(41 + dim)
    t_hi = max(p[:, 2].max() for p in paths)
    p0, p1 = _edges(rng, 6000, -50, 50, t_hi, dim)
    p1[:40] = p0[:40]                                 # zero-length edges (0/0 slopes)
    p1[40:80, 2] = p0[40:80, 2]                       # edges without duration (division by zero)
    with Context(dim) as ctx:
        ctx.polygons_set(polys, kinds=kinds, paths=paths)
        for rr in (0.5, 4.0):
            hit, first = ctx.edges_check(p0, p1, rr, kind=1)
            ohit, ofirst = oracle.edges_check_polygons(ps, p0, p1, rr)
            assert np.array_equal(hit, ohit) and np.array_equal(first, ofirst)
            assert 0.02 < hit.mean() < 0.98
        for j in (0, 5, m - 1):                       # explicitEdgeCheck(S, edge, ob): one obstacle
            hit, _ = ctx.edges_check(p0[:1500], p1[:1500], 2.0, kind=1, obstacle=j)
            one = oracle.PolygonSet([polys[j]], kinds=[kinds[j]], paths=[paths[j]])
            assert np.array_equal(hit, oracle.edges_check_polygons(one, p0[:1500], p1[:1500], 2.0)[0])
        pts = p0[:3000].copy()
        unsafe, clr = ctx.points_check(pts, 0.5, kind=1)
        exp = [oracle.point_check_polygons(ps, p, 0.5) for p in pts]
        assert np.array_equal(unsafe.astype(bool), np.array([e[0] for e in exp]))
        assert np.array_equal(clr, np.array([e[1] for e in exp]))
        assert 0.01 < unsafe.mean() < 0.9


def test_mixed_static_and_moving_with_inactive(oracle):
    rng = np.random.default_rng(5)
    polys, paths, kinds, active = [], [], [], []
    for i in range(24):
        c = rng.uniform(-40, 40, 2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, rng.integers(3, 6)))
        polys.append(c + np.c_[np.cos(ang), np.sin(ang)] * rng.uniform(2, 7))
        k = [3, 6, 7][i % 3]
        kinds.append(k)
        if k == 3:
            paths.append(None)
        else:
            rows = int(rng.integers(1, 9))            # a single-row path: never overlaps an edge in time
            t = np.sort(rng.uniform(0, 40, rows))
            paths.append(np.c_[rng.uniform(-30, 30, (rows, 2)), t])
        active.append(0 if i % 7 == 3 else 1)
    ps = oracle.PolygonSet(polys, kinds=kinds, active=active, paths=paths)
    p0, p1 = _edges(rng, 5000, -50, 50, 40, 3)
    with Context(3) as ctx:
        ctx.polygons_set(polys, kinds=kinds, active=active, paths=paths)
        hit, first = ctx.edges_check(p0, p1, 1.0, kind=1)
        ohit, ofirst = oracle.edges_check_polygons(ps, p0, p1, 1.0)
        assert np.array_equal(hit, ohit) and np.array_equal(first, ofirst)
        unsafe, clr = ctx.points_check(p0[:2500], 0.75, kind=1)
        exp = [oracle.point_check_polygons(ps, p, 0.75) for p in p0[:2500]]
        assert np.array_equal(unsafe.astype(bool), np.array([e[0] for e in exp]))
        assert np.array_equal(clr, np.array([e[1] for e in exp]))
        # a new path for one obstacle (kind 7's changeObstacleDirection) changes the answers accordingly
        paths2 = list(paths)
        paths2[1] = np.array([[0, 0, 0], [25, -10, 20], [25, 30, 40]], dtype=np.float64)
        ctx.polygon_paths_set(paths2)
        ps2 = oracle.PolygonSet(polys, kinds=kinds, active=active, paths=paths2)
        hit2, first2 = ctx.edges_check(p0, p1, 1.0, kind=1)
        ohit2, ofirst2 = oracle.edges_check_polygons(ps2, p0, p1, 1.0)
        assert np.array_equal(hit2, ohit2) and np.array_equal(first2, ofirst2)
        assert not np.array_equal(first, first2)


def test_hand_derived_cases_and_errors():
    sq = [[-1, -1], [1, -1], [1, 1], [-1, 1]]
    path = [[0, 0, 0], [10, 0, 10]]
    with Context(3) as ctx:
        ctx.polygons_set([sq], kinds=[6])
        with pytest.raises(_capi.RrtxError) as e:     # a moving obstacle needs its path
            ctx.edges_check([[0, 0, 0]], [[1, 1, 1]], 0.5, kind=1)
        assert e.value.code == _capi.RRTX_E_STATE and "no path" in str(e.value)
        ctx.polygons_set([sq], kinds=[6], paths=[path])
        p0 = np.array([[5, -5, 0], [5, 5, 10], [5, -5, 20], [0, 3, 0], [5, -5, 0]], dtype=np.float64)
        p1 = np.array([[5, 5, 10], [5, -5, 0], [5, 5, 30], [10, 3, 10], [5, 5, 2]], dtype=np.float64)
        hit, _ = ctx.edges_check(p0, p1, 0.1, kind=1)
        assert list(hit) == [1, 1, 0, 0, 0]           # M1, M1', M2, M3 of tests/test_oracle_kat.py
        hit, _ = ctx.edges_check(p0[2:4], p1[2:4], 6.0, kind=1)
        assert list(hit) == [0, 0]
        unsafe, clr = ctx.points_check([[5, 0, 5], [5, 0, 0], [10.5, 0, 100], [4.25, 0, 2.5]], 0.5, kind=1)
        assert list(unsafe) == [1, 0, 1, 0] and list(clr) == [0.0, 3.5, 0.0, 0.25]
        with pytest.raises(_capi.RrtxError):
            ctx.polygon_paths_set([path, path])       # count must match the polygon list
        with pytest.raises(_capi.RrtxError) as e:
            ctx.polygons_set([sq], kinds=[5])
        assert e.value.code == _capi.RRTX_E_INVALID
    with Context(4) as ctx:                           # Dubins + moving obstacles: refused, not guessed
        ctx.set_wrap(3, 2 * np.pi)
        ctx.polygons_set([sq], kinds=[7], paths=[path])
        with pytest.raises(_capi.RrtxError) as e:
            ctx.dubins_edges_check([[0, 0, 0, 0]], [[5, 5, 0, 1]], 1.0, 0.5)
        assert e.value.code == _capi.RRTX_E_STATE and "moving" in str(e.value)


def test_mirror_reads_time_obstacle_files(oracle, tmp_path):
    polys, paths = _env()
    f = tmp_path / "time_obs.txt"
    envio.write_time_obstacles(str(f), envio.TimeObstacleEnv(polys[::-1], np.full(len(polys), 3.0), paths[::-1]))
    back = envio.read_time_obstacles(str(f))
    assert all(np.allclose(a, b, atol=1e-6) for a, b in zip(back.polygons, polys[::-1]))
    S = drrt.CSpace(3, 0.0, [-50, -50, 0], [50, 50, 40], [0, 0, 0], [0, 0, 0])
    S.robotRadius = 0.5
    tree = drrt.KDTree(3)
    S.bind(tree)
    drrt.readTimeObstaclesFromfile(S, str(f), 1)
    obs = list(S.obstacles)
    assert len(obs) == len(polys) and all(o.kind == 6 for o in obs)
    ps = oracle.PolygonSet([o.polygon for o in obs], kinds=[6] * len(obs), paths=[o.path for o in obs])
    rng = np.random.default_rng(8)
    p0, p1 = _edges(rng, 400, -50, 50, 40, 3)
    edges = [drrt.newEdge(drrt.RRTNode(a), drrt.RRTNode(b)) for a, b in zip(p0, p1)]
    got = drrt.explicitEdgeChecks(S, edges)
    assert np.array_equal(np.asarray(got, dtype=np.uint8), oracle.edges_check_polygons(ps, p0, p1, 0.5)[0])
    u, c = drrt.explicitPointCheck(S, p0[0])
    assert (u, c) == oracle.point_check_polygons(ps, p0[0], 0.5)
    # the batched preamble picks the polygon list by itself when CSpace.obstacles holds Obstacles
    nodes = rng.uniform([-50, -50, 0], [50, 50, 40], (400, 3))
    for p in nodes:
        drrt.kdInsert(tree, drrt.RRTNode(p))
    samples = rng.uniform([-50, -50, 0], [50, 50, 40], (64, 3))
    out = drrt.extend_candidates(tree, S, samples, 25.0)
    q0, q1 = synth.candidate_edges(samples, nodes, out["offsets"], out["idx"])
    rh = oracle.edges_check_polygons(ps, q0, q1, 0.5)[0]
    k = len(out["idx"])
    assert k > 200 and np.array_equal(out["hit_out"], rh[:k]) and np.array_equal(out["hit_in"], rh[k:])
