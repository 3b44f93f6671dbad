"""Pins the CPU oracle (oracle/rrtx_oracle.c).

The reference holds no golden vectors or asserting tests for this path
(SURVEY.md 8(c)), and Julia is not available to run it, so the oracle is pinned
by (1) the known answers K1-K10 derived by hand from the reference source and
(2) the reference's own differential design: kd-tree search == naive scan
(R/kdTree_general.jl:1039-1148, commented testCase/testGhost).
"""
import math

import numpy as np
import pytest


def test_k1_euclid(oracle):
    assert oracle.euclid([0, 0, 0], [3, 4, 0]) == 5.0


def test_k2_range_inclusivity(oracle):
    t = oracle.KDTree(3)
    t.insert([0, 0, 0])
    t.insert([3, 4, 0])
    idx, key = t.within_range(5.0, [0, 0, 0])
    assert list(idx) == [0] and list(key) == [0.0]           # A: 5.0 < 5.0 is false
    idx, key = t.within_range(5.0, [3, 4, 0])
    assert sorted(idx) == [0, 1]                              # root: 5.0 <= 5.0
    assert dict(zip(idx, key)) == {0: 5.0, 1: 0.0}


def test_k3_to_k6_sphere_edge_quirks(oracle):
    sp, m = oracle.make_spheres([[0, 0, 0, 1.0]])
    # K3: t = dot/L (not L^2) pushes the foot point to p1 => no hit although the true distance is 1.4 < 1.5
    assert oracle.edge_check_spheres(sp, m, [-2, 1.4, 0], [2, 1.4, 0], 0.5) == (False, -1)
    assert oracle.distance_point_to_segment3([0, 0, 0], [-2, 1.4, 0], [2, 1.4, 0]) == math.sqrt(4 + 1.4 * 1.4)
    # K4: short edge, t clamps to 0
    assert oracle.edge_check_spheres(sp, m, [0.9, 0, 0], [1.4, 0, 0], 0.5) == (True, 0)
    assert oracle.distance_point_to_segment3([0, 0, 0], [0.9, 0, 0], [1.4, 0, 0]) == 0.9
    # K5: zero-length edge => 0/0 = NaN => comparison false => hit
    assert oracle.edge_check_spheres(sp, m, [10, 10, 10], [10, 10, 10], 0.5) == (True, 0)
    assert math.isnan(oracle.distance_point_to_segment3([0, 0, 0], [10, 10, 10], [10, 10, 10]))
    # K6: inactive obstacle never hits
    sp2, m2 = oracle.make_spheres([[0, 0, 0, 1.0]], active=[0])
    assert oracle.edge_check_spheres(sp2, m2, [0.9, 0, 0], [1.4, 0, 0], 0.5) == (False, -1)
    sp3, m3 = oracle.make_spheres([[0, 0, 0, 1.0]], life_span=[0.0])
    assert oracle.edge_check_spheres(sp3, m3, [0.9, 0, 0], [1.4, 0, 0], 0.5) == (False, -1)


def test_edge_direction_dependence(oracle):
    """the dot/L formula makes the result depend on the edge direction (why extend checks both)"""
    sp, m = oracle.make_spheres([[0, 0, 0, 1.0]])
    a, b = [-1.6, 1.45, 0], [3.0, 1.45, 0]
    d_ab = oracle.distance_point_to_segment3([0, 0, 0], a, b)
    d_ba = oracle.distance_point_to_segment3([0, 0, 0], b, a)
    assert d_ab != d_ba


def test_k7_polygon(oracle):
    ps = oracle.PolygonSet([[[0, 0], [1, 0], [1, 1], [0, 1]]])
    cr = ps.centre_radius()[0]
    assert cr[0] == 0.5 and cr[1] == 0.5 and cr[2] == math.sqrt(0.5)
    assert oracle.edge_check_polygons(ps, [-1, .5], [2, .5], 0.1) == (True, 0)
    assert oracle.edge_check_polygons(ps, [-1, 2], [2, 2], 0.1) == (False, -1)   # bounding-circle reject


def test_k8_point_segment(oracle):
    assert oracle.dist_sqrd_point_to_segment([1, 1], [0, 0], [2, 0]) == 1.0   # interior branch
    assert oracle.dist_sqrd_point_to_segment([0, 1], [0, 0], [2, 0]) == 1.0   # det == 0 -> start branch
    assert oracle.dist_sqrd_point_to_segment([3, 1], [0, 0], [2, 0]) == 2.0   # end branch


def test_segment_dist_branches(oracle):
    # crossing segments -> 0.0
    assert oracle.segment_dist_sqrd([0, 0], [2, 2], [0, 2], [2, 0]) == 0.0
    # near-vertical P with Q strictly on one side
    assert oracle.segment_dist_sqrd([0, 0], [0, 2], [1, 0], [1, 2]) == 1.0
    # parallel horizontal segments
    assert oracle.segment_dist_sqrd([0, 0], [2, 0], [0, 1], [2, 1]) == 1.0
    # touching at an end point: the reference's strict side test reports an intersection
    assert oracle.segment_dist_sqrd([0, 0], [2, 0], [1, 0], [1, 3]) == 0.0


def test_k11_segments_on_one_line_far_apart(oracle):
    """Hand-derived from R/DRRT.jl:1144-1202: with P = (5,0)-(6,0) and Q = (0,0)-(1,0) the slope of P is 0, both
    differences (m*(Q.x - PA.x) + PA.y) - Q.y are exactly 0, so "Q on one side of P" (strict > / <) fails; the same
    for P against Q; possibleIntersect stays true and the function returns 0.0 although the segments are 4 apart.
    On a vertical common line the first branch compares with >= / <= and does separate them: the distance is
    real.  A slanted common line behaves like the horizontal one whenever the differences round to 0 (here all
    four products are exact).  The device's segment rejection has to reproduce this (kernels_collide.hip, stage A)."""
    assert oracle.segment_dist_sqrd([5, 0], [6, 0], [0, 0], [1, 0]) == 0.0
    assert oracle.segment_dist_sqrd([0, 5], [0, 6], [0, 0], [0, 1]) == 16.0
    assert oracle.segment_dist_sqrd([8, 4], [12, 6], [0, 0], [2, 1]) == 0.0           # slope 1/2, exact
    # one unit off the common line: the side tests separate, the result is the end-point distance
    assert oracle.segment_dist_sqrd([5, 1], [6, 1], [0, 0], [1, 0]) == 17.0


def test_k12_point_in_polygon_ray_through_a_vertex(oracle):
    """Hand-derived from R/DRRT.jl:1009-1056: an edge is counted only if its ends lie STRICTLY on opposite sides of
    the ray's height.  Triangle (0,0), (2,-1), (2,1), point (-5, 0): the two edges at the vertex (0,0) are not
    counted (one end at the ray's height), the edge (2,-1)-(2,1) is, with both x right of the point: one crossing,
    odd, "inside" -- for a point 5 to the left of the polygon.  One ulp off that height the count is 2.  The
    explicit point check meets this whenever it evaluates such an obstacle, however far away it is, which is why a
    flag-only call cannot leave far obstacles out (kernels_collide.hip, points_polygons_kernel)."""
    tri = [[0, 0], [2, -1], [2, 1]]
    assert oracle.point_in_polygon([-5, 0], tri)
    assert not oracle.point_in_polygon([-5, np.nextafter(0.0, 1.0)], tri)
    assert not oracle.point_in_polygon([-5, 0.5], tri) and oracle.point_in_polygon([1.5, 0.5], tri)
    # a point on the far side of the same ray: no edge has both ends to its right
    assert not oracle.point_in_polygon([5, 0], tri)


def test_point_in_polygon(oracle):
    sq = [[0, 0], [1, 0], [1, 1], [0, 1]]
    assert oracle.point_in_polygon([.5, .5], sq)
    assert not oracle.point_in_polygon([1.5, .5], sq)
    assert not oracle.point_in_polygon([-.5, .5], sq)
    tri = [[0, 0], [4, 0], [0, 4]]
    assert oracle.point_in_polygon([1, 1], tri)
    assert not oracle.point_in_polygon([3, 3], tri)


def test_k9_ball_radius(oracle):
    assert abs(oracle.ball_radius(8, 80, 200000, 3) - 3.1497) < 1e-4
    assert abs(oracle.ball_radius(8, 80, 10000, 3) - 7.7837) < 1e-4
    assert abs(oracle.ball_radius(8, 80, 50000, 3) - 4.8029) < 1e-4
    assert abs(oracle.ball_radius(8, 80, 500000, 3) - 2.3774) < 1e-4
    assert oracle.ball_radius(10, 100, 50000, 4) == 10.0
    assert abs(oracle.ball_radius(10, 100, 500000, 4) - 7.1575) < 1e-4
    assert oracle.ball_radius(8, 80, 100, 3) == 8.0             # delta caps small trees


def test_k10_dubins_straight_ahead(oracle):
    """s=(0,0,.,0) -> g=(10,0,.,0), r=1: the x axis is tangent to all four circles, so rsl, rsr,
    lsr and lsl all have cost 10; evaluation order rsl first + strict `bestDist > len` keeps "rsl".
    (SURVEY.md's hand derivation predicted "rsr"; the inner tangent of irc/glc is the x axis too.)"""
    c, w, traj = oracle.dubins_steer([0, 0, 0, 0], [10, 0, 0, 0], 1.0)
    assert abs(c - 10.0) <= 1e-12
    assert w in ("rsl", "rsr", "lsr", "lsl")
    assert traj.shape == (4, 2)          # 1 + 2 + 1 rows: both arcs are single points
    assert np.allclose(traj[[0, -1]], [[0, 0], [10, 0]], atol=1e-12)


def test_dubins_known_shapes(oracle):
    # quarter-turn geometry: start heading +x at origin, goal at (2, 2) heading +y with r = 2: one left arc
    c, w, traj = oracle.dubins_steer([0, 0, 0, 0], [2, 2, 0, math.pi / 2], 2.0)
    assert abs(c - math.pi) < 1e-9 and "l" in w     # degenerate pieces tie; "rsl" is evaluated first
    # U-turn needs the curve-curve-curve family when the goal is close
    c2, w2, _ = oracle.dubins_steer([0, 0, 0, 0], [0, 0.5, 0, math.pi], 1.0)
    assert w2 in ("rlr", "lrl", "lsl", "rsr", "lsr", "rsl") and c2 > 0
    # symmetric mirror: left/right words swap, cost is equal
    ca, wa, _ = oracle.dubins_steer([0, 0, 0, 0.3], [6, 3, 0, 1.1], 1.0)
    cb, wb, _ = oracle.dubins_steer([0, 0, 0, -0.3], [6, -3, 0, -1.1], 1.0)
    assert abs(ca - cb) < 1e-9
    assert wa.translate(str.maketrans("lr", "rl")) == wb
    # trajectory sanity: consecutive arc samples are 0.1 rad apart on a circle of radius r_min
    _, _, traj = oracle.dubins_steer([0, 0, 0, 0], [2, 2, 0, math.pi / 2], 2.0)
    d = np.linalg.norm(np.diff(traj, axis=0), axis=1)
    chord = 2 * 2.0 * math.sin(0.05)
    assert (np.abs(d - chord) < 1e-9).sum() >= 14          # pi/2 of arc = 15 full 0.1-rad steps
    assert (d <= chord + 1e-9).all()


def test_julia_range_len(oracle):
    # literal-fallback branch of Julia's float range: len = round((stop-start)/step)+1, minus overshoot
    assert oracle.julia_range_len(0.0, 0.1, 0.35) == 4          # 0, .1, .2, .3
    assert oracle.julia_range_len(0.0, -0.1, -0.35) == 4
    assert oracle.julia_range_len(0.0, 0.1, -1.0) == 0
    assert oracle.julia_range_len(0.3, 0.1, 0.3) == 1
    assert oracle.julia_range_len(1.0, 0.1, 1.26) == 3          # 3.6 rounds to 4 -> overshoot -> 3


@pytest.mark.parametrize("d", [3, 4, 6])
def test_kd_equals_naive(oracle, d):
    """the reference's testCase design: kd search == naive scan (nearest, and range as a set with keys)"""
    rng = np.random.default_rng(100 + d)
    pts = rng.random((4000, d))
    t = oracle.KDTree(d)
    t.insert_many(pts)
    assert t.size == 4000
    for q in rng.random((200, d)):
        assert t.nearest(q) == t.nearest(q, naive=True)
        idx, key = t.within_range(0.25, q)
        nidx, nkey = t.range_naive(0.25, q)
        o = np.argsort(idx)
        assert np.array_equal(idx[o], nidx) and np.array_equal(key[o], nkey)


@pytest.mark.parametrize("d", [3, 4])
def test_kd_knearest_equals_naive(oracle, d):
    """kdFindKNearest vs kdFindKNearestNaive (the reference's own differential design,
    R/kdTree_general.jl:1064): same node set, same keys, for k >= 2."""
    rng = np.random.default_rng(300 + d)
    pts = rng.random((3000, d))
    t = oracle.KDTree(d)
    t.insert_many(pts)
    for k in (2, 3, 17, 64):
        for q in rng.random((40, d)):
            idx, key = t.knearest(k, q)
            nidx, nkey = t.knearest(k, q, naive=True)
            assert len(idx) == k
            o, no = np.argsort(idx), np.argsort(nidx)
            assert np.array_equal(idx[o], nidx[no]) and np.array_equal(key[o], nkey[no])
            assert key.max() == key[0]                      # heap order: the farthest is on top


def test_kd_knearest_seed_quirks(oracle):
    # the heap starts with root + an Inf dummy (R/kdTree_general.jl:699-706): k = 1 hands back the
    # TWO nearest nodes; a tree smaller than k hands back all of it; wrapped spaces raise (:711-713)
    t = oracle.KDTree(2)
    for p in [[0, 0], [1, 0], [3, 0], [7, 0]]:
        t.insert(p)
    idx, key = t.knearest(1, [2.9, 0])
    assert sorted(idx) == [1, 2] and sorted(key) == [pytest.approx(0.1), pytest.approx(1.9)]
    idx, _ = t.knearest(1, [-1.0, 0])
    assert sorted(idx) == [0, 1]
    idx, key = t.knearest(9, [0, 0])
    assert sorted(idx) == [0, 1, 2, 3] and sorted(key) == [0.0, 1.0, 3.0, 7.0]
    one = oracle.KDTree(2)
    one.insert([5, 5])
    idx, key = one.knearest(3, [5, 6])
    assert list(idx) == [0] and list(key) == [1.0]
    w = oracle.KDTree(2, wraps=[1], wrap_points=[1.0])
    w.insert([.5, .5])
    with pytest.raises(RuntimeError):
        w.knearest(2, [.1, .1])


def test_kd_range_list_order_is_reverse_discovery(oracle):
    # JlistPush inserts at the front: the root (added first when within range) is the LAST element
    t = oracle.KDTree(2)
    for p in [[.5, .5], [.25, .5], [.75, .5], [.1, .1]]:
        t.insert(p)
    idx, _ = t.within_range(10.0, [.5, .5])
    assert idx[-1] == 0 and sorted(idx) == [0, 1, 2, 3]


def test_find_more_within_range_keeps_first_key(oracle):
    t = oracle.KDTree(2)
    for p in [[0, 0], [1, 0], [3, 0]]:
        t.insert(p)
    idx, key = t.within_range(1.5, [0, 0], more=[(2.5, [2, 0])])
    got = dict(zip(idx, key))
    assert got == {0: 0.0, 1: 1.0, 2: 1.0}     # node 1 keeps the key of its first discovery


def test_ghost_points_single_wrap(oracle):
    """testGhost design: R^2 x S^1; the ghost of theta is theta + 2pi (theta < pi) or theta - 2pi"""
    two_pi = 2 * math.pi
    t = oracle.KDTree(4, wraps=[3], wrap_points=[two_pi])
    t.insert([0, 0, 0, 1.0])
    g = t.ghost_points([1, 2, 0, 0.5], 10.0)
    assert g.shape == (1, 4) and g[0, 3] == 0.5 + two_pi and list(g[0, :3]) == [1, 2, 0]
    g = t.ghost_points([1, 2, 0, 6.0], 10.0)
    assert g[0, 3] == 6.0 - two_pi
    # skipped when the unwrapped point closest to the ghost is farther than bestDist
    assert t.ghost_points([1, 2, 0, 3.0], 1.0).shape[0] == 0
    assert t.ghost_points([1, 2, 0, 0.5], 0.4).shape[0] == 0
    assert t.ghost_points([1, 2, 0, 0.5], 0.6).shape[0] == 1


def test_ghost_points_two_wraps_order(oracle):
    t = oracle.KDTree(3, wraps=[0, 2], wrap_points=[1.0, 1.0])
    t.insert([.5, .5, .5])
    g = t.ghost_points([.1, .5, .9], 10.0)
    # iteration order: last wrapped dim first, then the first, then both
    assert np.allclose(g, [[.1, .5, -.1], [1.1, .5, .9], [1.1, .5, -.1]])


def test_wrapped_range_equals_naive(oracle):
    two_pi = 2 * math.pi
    rng = np.random.default_rng(9)
    pts = np.concatenate([rng.uniform(-5, 5, (3000, 2)), np.zeros((3000, 1)), rng.uniform(0, two_pi, (3000, 1))], 1)
    t = oracle.KDTree(4, wraps=[3], wrap_points=[two_pi])
    t.insert_many(pts)
    for q in pts[rng.integers(0, 3000, 100)] + 0.01:
        for r in (1.0, 4.0):
            idx, key = t.within_range(r, q)
            nidx, nkey = t.range_naive(r, q)
            o = np.argsort(idx)
            assert np.array_equal(idx[o], nidx) and np.array_equal(key[o], nkey)


def test_point_checks(oracle):
    sp, m = oracle.make_spheres([[0, 0, 0, 1.0], [5, 0, 0, 2.0]])
    assert oracle.point_check_spheres(sp, m, [0.5, 0, 0], 0.5) == (True, 0.0)        # inside (quickCheck)
    assert oracle.point_check_spheres(sp, m, [1.2, 0, 0], 0.5) == (True, 0.0)        # robot radius overlaps
    unsafe, clr = oracle.point_check_spheres(sp, m, [2.5, 0, 0], 0.5)
    assert not unsafe and clr == 0.0 + (2.5 - 0.5 - 2.0)                               # sphere 2: 2.5-0.5-2
    unsafe, clr = oracle.point_check_spheres(sp, m, [0, 10, 0], 0.5)
    assert not unsafe and clr == (10.0 - 0.5) - 1.0
    assert oracle.point_check_spheres(sp, 0, [0, 0, 0], 0.5) == (False, float("inf"))


def test_moving_obstacle_edge_kats(oracle):
    """Kinds 6 / 7 (R/DRRT_Q.jl:1699-1771), derived by hand.  Obstacle: unit square about the origin
    (ctor centre (0,0), radius sqrt 2) sliding along +x at speed 1: path (0,0,t=0) -> (10,0,t=10).
    M1  robot (5,-5,t=0) -> (5,5,t=10): slopes (0,1) vs (1,0); T_c = (5 + 5)/2 = 5, both centres at
        (5,0) => hit.  M1' the same edge given end-first is re-ordered past->future => hit.
    M2  the same crossing 20 time units later lies after the path's last row: firstObsInd = 2,
        lastObsInd = min(3, 2) = 2 => "does not overlap in time" => no hit, although the obstacle is
        assumed to rest at (10,0) and a robot-radius of 6 would reach it (reference behaviour).
    M3  robot parallel to the obstacle, 3 apart: slopes equal => 0/0 = NaN => every comparison is
        false => no hit for sum of radii <= 3 ... and also none above it (NaN distance): a quirk."""
    sq = [[-1, -1], [1, -1], [1, 1], [-1, 1]]
    path = [[0, 0, 0], [10, 0, 10]]
    ps = oracle.PolygonSet([sq], kinds=[6], paths=[path])
    chk = lambda a, b, rr: int(oracle.edges_check_polygons(ps, np.array([a], float), np.array([b], float), rr)[0][0])
    assert chk([5, -5, 0], [5, 5, 10], 0.1) == 1
    assert chk([5, 5, 10], [5, -5, 0], 0.1) == 1
    assert chk([5, -5, 20], [5, 5, 30], 6.0) == 0
    assert chk([0, 3, 0], [10, 3, 10], 0.5) == 0
    assert chk([0, 3, 0], [10, 3, 10], 5.0) == 0
    # a miss in space-time: the robot crosses x = 5 when the obstacle is still at x = 1
    assert chk([5, -5, 0], [5, 5, 2], 0.1) == 0
    ps7 = oracle.PolygonSet([sq], kinds=[7], paths=[path])
    assert int(oracle.edges_check_polygons(ps7, np.array([[5., -5, 0]]), np.array([[5., 5, 10]]), 0.1)[0][0]) == 1


def test_moving_obstacle_point_kats(oracle):
    """findTransformObsToTimeOfPoint (R/DRRT_Q.jl:1367-1391) + the kind 6 branches of quickCheck2D /
    explicitPointCheck2D (R/DRRT.jl:1289-1305, 1395-1420); same obstacle as above."""
    sq = [[-1, -1], [1, -1], [1, 1], [-1, 1]]
    ps = oracle.PolygonSet([sq], kinds=[6], paths=[[[0, 0, 0], [10, 0, 10]]])
    assert oracle.point_check_polygons(ps, [5, 0, 5], 0.5) == (True, 0.0)          # square is at x = 5
    unsafe, clr = oracle.point_check_polygons(ps, [5, 0, 0], 0.5)                    # square still at 0
    assert not unsafe and clr == 3.5
    assert oracle.point_check_polygons(ps, [10.5, 0, 100], 0.5) == (True, 0.0)       # rests at path[end]
    assert oracle.point_check_polygons(ps, [0.5, 0, -3], 0.5) == (True, 0.0)         # before: path[1]
    unsafe, clr = oracle.point_check_polygons(ps, [2.5 + 1 + 0.75, 0, 2.5], 0.5)     # 0.75 off the edge
    assert not unsafe and clr == 0.25


def test_reference_env_files_as_inputs(oracle):
    """the two environment fixtures the reference ships are inputs, not expected outputs; the parsed
    copies under tests/golden must load and have the documented shape"""
    import json, os
    g = os.path.join(os.path.dirname(__file__), "golden")
    env = json.load(open(os.path.join(g, "env_inputs.json")))
    b2 = np.array(env["building2_spheres"])
    assert b2.shape == (31, 4) and (b2[:, 3] == 3.5).all()
    rs = env["rand_Static_polygons"]
    assert len(rs) == 35 and all(len(p) == 4 for p in rs)


# ---- Dubins in a space with time (R/DRRT_DubinsEdge_functions.jl:115-121, 660-697, 750-774), hand-derived ----
def test_T1_dubins_time_straight_ahead(oracle):
    """s = (0, 0, t=10, theta=0) -> g = (10, 0, t=0, theta=0), r_min = 1: Wdist = 10 (K10), dt = 10 - 0, so
    edge.dist = sqrt(10^2 + 10^2), edge.velocity = 10 / 10 = 1; trajectory: row 1 carries the start time, the
    last row IS the end node (x, y, t), rows between carry start time - distance walked / velocity."""
    dist, wdist, vel, word, traj = oracle.dubins_steer_time([0, 0, 10, 0], [10, 0, 0, 0], 1.0)
    assert abs(wdist - 10.0) < 1e-12 and abs(dist - np.sqrt(200.0)) < 1e-12 and abs(vel - 1.0) < 1e-12
    assert traj.shape[1] == 3 and traj[0, 2] == 10.0
    assert np.array_equal(traj[-1], [10.0, 0.0, 0.0])            # made exact, :696
    walked = np.concatenate([[0.0], np.cumsum(np.hypot(*np.diff(traj[:, :2], axis=0).T))])
    assert np.allclose(traj[1:-1, 2], 10.0 - walked[1:-1] / vel, atol=1e-12)
    assert (np.diff(traj[:, 2]) <= 0).all()                      # reverse time: the robot moves toward t = 0
    # no route (word xxx never happens for finite input, but Wdist = Inf must stay Inf): dist of an Inf cost
    # is Inf; here: the non-time cost equals Wdist
    c, w, t2 = oracle.dubins_steer([0, 0, 10, 0], [10, 0, 0, 0], 1.0)
    assert c == wdist and w == word and np.array_equal(t2[:-1], traj[:-1, :2])


def test_T2_dubins_time_valid_move(oracle):
    """validMove: start time > end time AND dubinsMinVelocity <= velocity <= dubinsMaxVelocity (:120)"""
    s, g = [0, 0, 10, 0], [10, 0, 0, 0]
    assert not oracle.dubins_valid_move_time(s, g, 1.0, 5.0, 30.0)          # too slow
    assert oracle.dubins_valid_move_time(s, g, 10.0, 5.0, 30.0)
    assert oracle.dubins_valid_move_time(s, g, 5.0, 5.0, 30.0) and oracle.dubins_valid_move_time(s, g, 30.0, 5.0, 30.0)
    assert not oracle.dubins_valid_move_time(s, g, 30.000001, 5.0, 30.0)
    assert not oracle.dubins_valid_move_time(g, s, 10.0, 5.0, 30.0)         # forward in time: never
    assert not oracle.dubins_valid_move_time(s, [10, 0, 10, 0], 10.0, 5.0, 30.0)   # no duration
    assert not oracle.dubins_valid_move_time(s, g, float("nan"), 5.0, 30.0)


def test_T3_dubins_time_moving_obstacle(oracle):
    """A unit square (bounding radius sqrt(.5)) drifting up the line x = 5 crosses the x axis at time 5; the
    robot of T1 is at x = 5 at time 5 (x = 10 - t): the centres meet -> hit.  The same square held below
    y = -4 until t = 8 and only then rushed across: in [8, 10] the robot is at x <= 2, at least 3 from the
    line x = 5, farther than robotRadius + sqrt(.5) = 1.21 (stage 2) and than the inflated 3.21 of the
    chord test when the square passes y = 0 at t = 8.9 (robot at x = 1.1: 3.9 away) -> no hit."""
    sq = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]]) + [5.0, -5.0]
    s, g = [0, 0, 10, 0], [10, 0, 0, 0]
    _, _, _, _, traj = oracle.dubins_steer_time(s, g, 1.0)
    meet = oracle.PolygonSet([sq], kinds=[6], paths=[np.array([[0, 0, 0.0], [0, 10, 10.0]])])
    assert oracle.dubins_edge_check_polygons_time(meet, s, g, traj, 0.5, 1.0) == (True, 0)
    late = oracle.PolygonSet([sq], kinds=[6], paths=[np.array([[0, 0, 0.0], [0, 1, 8.0], [0, 10, 10.0]])])
    assert oracle.dubins_edge_check_polygons_time(late, s, g, traj, 0.5, 1.0) == (False, -1)
    # a static polygon sitting on the route is hit whatever the times are
    static = oracle.PolygonSet([sq + [0.0, 5.0]])
    assert oracle.dubins_edge_check_polygons_time(static, s, g, traj, 0.5, 1.0) == (True, 0)


def test_T4_time_column_piecewise_vs_running_sum(oracle):
    """The time column of edge.trajectory: the reference adds the straight pieces between stored rows up one by one
    (R/DRRT_DubinsEdge_functions.jl:691-695; oracle default); the HIP kernels take (distance at the first row of the
    piece) + k x (the piece's first chord) (oracle piecewise=True, what the device is compared with bit for bit).
    Same x / y rows, time stamps equal to 1e-12 relative, first and last rows exact in both."""
    rng = np.random.default_rng(3)
    worst = 0.0
    for _ in range(400):
        s = np.r_[rng.uniform(-20, 20, 2), rng.uniform(10, 30), rng.uniform(0, 2 * math.pi)]
        g = np.r_[s[:2] + rng.normal(0, 8, 2), s[2] - rng.uniform(0.1, 5.0), rng.uniform(0, 2 * math.pi)]
        a = oracle.dubins_steer_time(s, g, 2.0)
        b = oracle.dubins_steer_time(s, g, 2.0, piecewise=True)
        assert a[:4] == b[:4] and a[4].shape == b[4].shape
        assert np.array_equal(a[4][:, :2], b[4][:, :2])
        assert np.array_equal(a[4][0], b[4][0]) and np.array_equal(a[4][-1], b[4][-1]) and np.array_equal(b[4][-1], g[:3])
        worst = max(worst, float(np.abs(a[4][:, 2] - b[4][:, 2]).max() / max(1.0, abs(s[2]))))
    assert 0.0 < worst < 1e-12


def test_K13_conflict_nodes_dubins_query(oracle):
    """findPointsInConflictWithObstacle(::Obstacle) in the Dubins space (R/DRRT.jl:3061-3065): query [x y 0.0 pi],
    range robotRadius + delta + ob.radius + pi, KDdist over all four coordinates, the root taken with <=.
    Unit square at the origin: centre (.5, .5), radius sqrt(.5); robotRadius .5, delta 1 => range = 2.2071 + pi."""
    rng_ = 0.5 + 1.0 + math.sqrt(0.5) + math.pi
    sq = [[0, 0], [1, 0], [1, 1], [0, 1]]
    ps = oracle.PolygonSet([sq])
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2 * math.pi])
    nodes = [
        [0.5 + rng_, 0.5, 0.0, math.pi],       # 0, the root, exactly at the range: in (<=)
        [0.5, 0.5, 0.0, 0.0],                   # 1: only the heading differs, by pi: in (pi < range)
        [0.5 + 4.0, 0.5, 0.0, math.pi],         # 2: 4.0 < range: in
        [0.5 + 5.4, 0.5, 0.0, math.pi],         # 3: 5.4 > range: out
        [0.5 + rng_, 0.5, 0.0, math.pi],        # 4: the root's twin, not the root: exactly at the range is OUT (<)
        [0.5 + 4.5, 0.5, 0.0, 6.2],             # 5: sqrt(4.5^2 + (6.2 - pi)^2) = 5.44 > range: out (the ghost at -pi is farther)
        [0.5, 0.5 - 3.0, 0.0, 1.0],             # 6: sqrt(9 + (pi - 1)^2) = 3.69: in
    ]
    for p in nodes:
        tree.insert(np.array(p))
    d0 = math.sqrt((((rng_ * rng_) + 0.0) + 0.0) + 0.0)
    assert d0 == rng_                                        # the construction really puts nodes 0 and 4 AT the range
    got = sorted(oracle.points_in_conflict_polygon(tree, ps, 0, 0.5, 1.0, False, True).tolist())
    assert got == [0, 1, 2, 6]
    with pytest.raises(RuntimeError):                        # a static obstacle in a space with time: the reference raises (:3067)
        oracle.points_in_conflict_polygon(tree, ps, 0, 0.5, 1.0, True, True)


def test_K14_conflict_nodes_moving_obstacle_path_queries(oracle):
    """kinds 6 / 7 (R/DRRT.jl:3070-3118): one query per path segment i -> i + 1 at [ob.position 0.0] + (path[i] +
    path[i+1]) / 2 (x, y and time) with range base + |path[i] - path[i+1]| / 2, accumulated; a one-row path gives one
    query at [ob.position 0.0] + path[1] with range base.  Euclidean space with time (d = 3: [x y t])."""
    sq = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])          # centre (0, 0), radius sqrt(.5)
    base = 0.5 + 1.0 + math.sqrt(0.5)                                            # 2.2071
    path = np.array([[0.0, 0.0, 0.0], [2.0, 0.0, 10.0], [2.0, 4.0, 20.0]])
    # query 1 at (1, 0, 5), range base + sqrt(104) / 2 = 7.306; query 2 at (2, 2, 15), range base + sqrt(116) / 2 = 7.592
    ps = oracle.PolygonSet([sq], kinds=[6], paths=[path])
    tree = oracle.KDTree(3)
    nodes = [
        [50.0, 50.0, 50.0],     # 0 root, far: out
        [1.0, 0.0, 5.0],        # 1 at query 1: in
        [1.0, 7.0, 5.0],        # 2: 7.0 from query 1: in
        [1.0, 7.4, 5.0],        # 3: 7.4 > 7.306 from query 1; from query 2: sqrt(1 + 5.4^2 + 100) = 11.4: out
        [2.0, 2.0, 22.5],       # 4: 7.5 from query 2 only: in
        [2.0, 2.0, 22.7],       # 5: 7.7 > 7.592: out
        [1.5, 1.0, 10.0],       # 6: within both (5.1 and 5.1): in once
    ]
    for p in nodes:
        tree.insert(np.array(p))
    got = oracle.points_in_conflict_polygon(tree, ps, 0, 0.5, 1.0, True, False).tolist()
    assert sorted(got) == [1, 2, 4, 6] and len(got) == 4          # node 6 is in the list once (inHeap)
    one = oracle.PolygonSet([sq], kinds=[7], paths=[np.array([[3.0, 0.0, 7.0]])])
    tree2 = oracle.KDTree(3)
    for p in ([40.0, 0.0, 0.0], [3.0, 0.0, 7.0], [3.0, 2.2, 7.0], [3.0, 2.21, 7.0]):
        tree2.insert(np.array(p))
    assert sorted(oracle.points_in_conflict_polygon(tree2, one, 0, 0.5, 1.0, True, False).tolist()) == [1, 2]
