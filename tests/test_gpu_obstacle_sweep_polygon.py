"""The obstacle sweeps of the polygon / Dubins space (the row the round-2 verdict added as A14b):
findPointsInConflictWithObstacle(::Obstacle) with the Dubins query [x y 0.0 pi] / range + pi and the per-path-segment
queries of kinds 6 / 7 (R/DRRT.jl:3048-3125), then the edge loops of addNewObstacle (:3127-3200) and removeObstacle
(:3202-3290) with explicitEdgeCheck(S, edge, ob) for SimpleEdge and DubinsEdge (R/DRRT_DubinsEdge_functions.jl:750-774),
through rrtx_obstacle_sweep_polygon against the oracle's restatement (oracle.add_new_obstacle_edges /
remove_obstacle_edges over the same mirror of directed edges): edge ids equal, one for one.

Obstacles: the reference's own discoverable set environments/rand_Disc_3.txt (tests/golden/env_inputs.json), the
moving polygons of rand_StaticTime_7.txt, and synth.dynamic_polygons (BASELINE config 5's recipe)."""
import json
import math
import os

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, synth
from rrtqx_3d_amd._capi import RrtxError
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
RR, DELTA = 0.5, 8.0
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _env():
    return json.load(open(os.path.join(G, "env_inputs.json")))


def _graph(oracle, tree, pts, r, rng, n_long=200):
    """a mirror like the planner's: both directed edges of every pair within r (what extend() links), a few long
    "parent" edges, some out-edges of the root and a zero-length edge"""
    es, ee = [], []
    for i in range(len(pts)):
        idx, _ = tree.within_range(r, pts[i])
        for j in np.sort(idx):
            if j != i:
                es.append(i); ee.append(int(j))
    es += rng.integers(0, len(pts), n_long).tolist()
    ee += rng.integers(0, len(pts), n_long).tolist()
    es += [0, 0, 0, 7]
    ee += [1, 2, 3, 7]
    return np.array(es, dtype=np.int32), np.array(ee, dtype=np.int32)


def test_simple_edges_discoverable_polygons(oracle):
    """SimpleEdge, the 2-D Euclidean space (a dim = 3 tree at z = 0) with the reference's rand_Disc_3.txt polygons:
    every obstacle appears (add), the blocked edges are recorded in the mirror, then every obstacle expires (remove)
    with the others still in use -- edges shared with a neighbouring obstacle must stay blocked."""
    env = _env()
    polys = [np.array(p) for p in env["rand_Disc_3_polygons"]]
    m = len(polys)
    rng = np.random.default_rng(5)
    n = 2500
    pts = np.c_[rng.uniform(-20, 20, (n, 2)), np.zeros(n)]
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    es, ee = _graph(oracle, tree, pts, 2.5, rng)
    active = np.ones(m, dtype=np.uint8)
    active[[4, 30]] = 0
    ps = oracle.PolygonSet(polys, active=active)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, active=active)
        assert ctx.graph_edges_append(es, ee) == 0
        blocked = np.zeros(len(es), dtype=bool)
        total = 0
        for j in range(m):
            got = ctx.obstacle_sweep_polygon(j, RR, DELTA, cap=8)                      # small capacity: two-call path
            want = oracle.add_new_obstacle_edges(tree, pts, es, ee, ps, j, RR, DELTA, dubins=False)
            assert np.array_equal(got, want), j
            blocked[got] = True
            total += len(got)
        assert total > 2000 and len(ctx.obstacle_sweep_polygon(4, RR, DELTA)) == 0    # an obstacle not in use
        ctx.graph_edges_block(np.nonzero(blocked)[0].astype(np.int32))
        dist = np.where(blocked, np.inf, 1.0)
        freed = 0
        for j in range(0, m, 3):
            got = ctx.obstacle_sweep_polygon(j, RR, DELTA, remove=True, cap=8)
            want = oracle.remove_obstacle_edges(tree, pts, es, ee, dist, ps, j, RR, DELTA, dubins=False)
            assert np.array_equal(got, want), j
            freed += len(got)
            # what the add loop found for j minus what it frees is held by another obstacle
            mine = ctx.obstacle_sweep_polygon(j, RR, DELTA)
            assert set(got.tolist()) <= set(mine.tolist())
        assert freed > 100
        with pytest.raises(RrtxError):
            ctx.obstacle_sweep_polygon(m, RR, DELTA)


def test_simple_edges_tree_off_the_plane_and_root_rule(oracle):
    """nodes with z != 0 (the query point is [x y 0.0]: KDdist runs over all coordinates) and a root exactly at the
    search range: kdFindWithinRange takes the root with <= (R/kdTree_general.jl:896), one ulp farther it is out"""
    rng = np.random.default_rng(8)
    n = 1500
    sq = np.array([[-2.0, -2.0], [2.0, -2.0], [2.0, 2.0], [-2.0, 2.0]])          # centre (0, 0), radius sqrt(8)
    rad = math.sqrt(8.0)
    rng_range = (RR + DELTA) + rad
    for root_x, expect_root in ((rng_range, True), (rng_range * (1.0 + 1e-12), False)):
        pts = np.c_[rng.uniform(-15, 15, (n, 2)), rng.uniform(-3, 3, n)]
        pts[0] = [root_x, 0.0, 0.0]
        pts[1] = [0.5, 0.25, 0.0]                                                  # the root's out-edge ends inside the square
        d_root = math.sqrt((root_x * root_x + 0.0) + 0.0)
        assert (d_root == rng_range) if expect_root else (d_root > rng_range)
        tree = oracle.KDTree(3)
        tree.insert_many(pts)
        es, ee = _graph(oracle, tree, pts, 3.0, rng)
        ps = oracle.PolygonSet([sq])
        with Context(3) as ctx:
            ctx.nodes_append(pts)
            ctx.polygons_set([sq])
            ctx.graph_edges_append(es, ee)
            got = ctx.obstacle_sweep_polygon(0, RR, DELTA)
            want = oracle.add_new_obstacle_edges(tree, pts, es, ee, ps, 0, RR, DELTA, dubins=False)
            assert np.array_equal(got, want) and len(got) > 20
            root_edge = int(np.nonzero((es == 0) & (ee == 1))[0][0])
            assert (root_edge in got.tolist()) == expect_root


def _dubins_tree(oracle, rng, n, span, with_time=False):
    pts = np.c_[rng.uniform(-span, span, (n, 2)), np.zeros(n), rng.uniform(0, 2 * math.pi, n)]
    if with_time:
        pts[:, 2] = rng.uniform(synth.T_MIN, synth.T_MAX, n)
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    return pts, tree


def test_dubins_edges_static_polygons(oracle):
    """DubinsEdge in [x y 0 theta] (theta wrapped): the query [x y 0.0 pi] with range + pi reaches every heading (and
    its ghost at theta = -pi is searched), the two-stage Dubins check runs against the ONE obstacle; add, block,
    remove against the oracle; rrtx_dubins_edges_check_obstacle on the same edges."""
    env = _env()
    polys = [np.array(p) for p in env["rand_Disc_3_polygons"]][:40]
    m = len(polys)
    rng = np.random.default_rng(11)
    pts, tree = _dubins_tree(oracle, rng, 1400, 20.0)
    es, ee = _graph(oracle, tree, pts, 4.0, rng, n_long=60)
    r_min = 1.0
    active = np.ones(m, dtype=np.uint8)
    active[9] = 0
    ps = oracle.PolygonSet(polys, active=active)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, active=active)
        ctx.graph_edges_append(es, ee)
        cost, _ = ctx.dubins_steer(pts[es], pts[ee], r_min)
        ctx.graph_edges_set_dist(0, cost)
        blocked = np.zeros(len(es), dtype=bool)
        total = 0
        for j in range(0, m, 2):
            got = ctx.obstacle_sweep_polygon(j, RR, DELTA, r_min=r_min, cap=8)
            want = oracle.add_new_obstacle_edges(tree, pts, es, ee, ps, j, RR, DELTA, dubins=True, r_min=r_min)
            assert np.array_equal(got, want), j
            blocked[got] = True
            total += len(got)
            if j % 10 == 0:     # the per-edge entry point (what the Julia shim's explicitEdgeCheck(S, edge, ob) calls)
                sel = rng.choice(len(es), 400, replace=False)
                h = ctx.dubins_edges_check_obstacle(pts[es[sel]], pts[ee[sel]], r_min, RR, j)
                ref = [oracle.explicit_edge_check_obstacle(ps, j, pts[es[e]], pts[ee[e]], RR, True, r_min) for e in sel]
                assert np.array_equal(h.astype(bool), np.array(ref))
        assert total > 500 and not ctx.dubins_edges_check_obstacle(pts[es[:50]], pts[ee[:50]], r_min, RR, 9).any()
        ctx.graph_edges_block(np.nonzero(blocked)[0].astype(np.int32))
        dist = np.where(blocked, np.inf, cost)
        freed = 0
        for j in range(0, m, 4):
            got = ctx.obstacle_sweep_polygon(j, RR, DELTA, r_min=r_min, remove=True)
            want = oracle.remove_obstacle_edges(tree, pts, es, ee, dist, ps, j, RR, DELTA, dubins=True, r_min=r_min)
            assert np.array_equal(got, want), j
            freed += len(got)
        assert freed > 50


def test_dubins_edges_with_time_moving_obstacles(oracle):
    """BASELINE config 5's space: DubinsEdge in [x y t theta] with obstacles that move in time (kinds 6 / 7) and
    static ones.  A moving obstacle is swept with one query per path segment (accumulated), its edges are checked at
    the pieces' time stamps; a static obstacle in a space with time is the reference's error (R/DRRT.jl:3067)."""
    env = _env()
    mv = [np.array(p) for p in env["rand_StaticTime_7_polygons"]][:6]
    mv_paths = [np.array(p) for p in env["rand_StaticTime_7_paths"]][:6]
    polys, kinds, paths, active, hidden = synth.dynamic_polygons(24)
    polys = mv + polys
    kinds = [6, 7, 6, 7, 6, 7] + list(kinds)
    paths = mv_paths + list(paths)
    m = len(polys)
    active = np.ones(m, dtype=np.uint8)
    rng = np.random.default_rng(13)
    pts, tree = _dubins_tree(oracle, rng, 900, 30.0, with_time=True)
    pts[:, 2] = rng.uniform(0.0, 30.0, len(pts))
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    es, ee = _graph(oracle, tree, pts, 7.0, rng, n_long=60)
    keep = pts[es, 2] > pts[ee, 2]                           # planning runs in reverse time: start later than end
    es, ee = es[keep], ee[keep]
    r_min = synth.R_MIN_TIME
    ps = oracle.PolygonSet(polys, kinds=kinds, paths=paths, active=active)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=active)
        ctx.set_space_has_time(True)
        ctx.graph_edges_append(es, ee)
        moving = [j for j in range(m) if kinds[j] in (6, 7)]
        static = [j for j in range(m) if kinds[j] == 3]
        with pytest.raises(RrtxError):
            ctx.obstacle_sweep_polygon(static[0], RR, DELTA, r_min=r_min)
        with pytest.raises(RuntimeError):
            oracle.points_in_conflict_polygon(tree, ps, static[0], RR, DELTA, True, True)
        blocked = np.zeros(len(es), dtype=bool)
        total = 0
        for j in moving:
            got = ctx.obstacle_sweep_polygon(j, RR, DELTA, r_min=r_min)
            want = oracle.add_new_obstacle_edges(tree, pts, es, ee, ps, j, RR, DELTA, dubins=True, r_min=r_min, has_time=True)
            assert np.array_equal(got, want), j
            blocked[got] = True
            total += len(got)
        assert total > 100
        ctx.graph_edges_block(np.nonzero(blocked)[0].astype(np.int32))
        dist = np.where(blocked, np.inf, 1.0)
        freed = 0
        for j in moving[:5]:
            got = ctx.obstacle_sweep_polygon(j, RR, DELTA, r_min=r_min, remove=True)
            want = oracle.remove_obstacle_edges(tree, pts, es, ee, dist, ps, j, RR, DELTA, dubins=True, r_min=r_min, has_time=True)
            assert np.array_equal(got, want), j
            freed += len(got)
        assert freed > 0


def test_sweep_queries_match_the_kd_tree_node_sets(oracle):
    """the node sets alone: node i's one mirrored edge is i -> i + 1, and an obstacle that covers the world (a ball,
    kind 1) makes every edge collide -- so the returned ids ARE findPointsInConflictWithObstacle's list, for the
    Dubins query with its ghost."""
    rng = np.random.default_rng(21)
    pts, tree = _dubins_tree(oracle, rng, 3000, 50.0)
    loop = np.arange(len(pts), dtype=np.int32)
    big = np.array([[-500.0, -500.0], [500.0, -500.0], [500.0, 500.0], [-500.0, 500.0]])
    small = np.array([[10.0, 10.0], [14.0, 10.0], [14.0, 13.0]])
    ps = oracle.PolygonSet([small, big], kinds=[3, 1])
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set([small, big], kinds=[3, 1])
        ctx.graph_edges_append(loop, (loop + 1) % len(pts))
        # obstacle 0 only selects the nodes; the check against it is what decides -- use the ball (1) for "all collide"
        for delta in (2.0, 8.0, 30.0):
            want = np.sort(oracle.points_in_conflict_polygon(tree, ps, 1, RR, delta, False, True))
            got = ctx.obstacle_sweep_polygon(1, RR, delta, r_min=1.0)
            assert np.array_equal(got, want)
        nodes0 = np.sort(oracle.points_in_conflict_polygon(tree, ps, 0, RR, 3.0, False, True))
        assert 0 < len(nodes0) < len(pts)
        th = pts[nodes0, 3]
        assert th.min() < 0.5 and th.max() > 5.5          # headings on both sides of the wrap are in the list
