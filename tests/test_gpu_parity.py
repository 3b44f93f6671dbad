"""GPU parity tests: the HIP path (through the C-ABI of include/rrtx.h) against
the CPU oracle on the same seeded inputs.

Bar (north_star): bit-exact neighbour indices / stored distances and collision
booleans; Dubins costs, words, row counts and booleans too since round 3 (device and oracle compile the one
deterministic sin / cos / atan2 / acos of include/rrtx_detmath.h); what stays un-pinnable is oracle <-> Julia's libm.
"""
import math

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu

ROBOT_RADIUS = 0.5      # R/experimentsForRRTQX.jl:38


@pytest.fixture(scope="module")
def small3(oracle):
    """N=10k nodes (config C2 scale), oracle kd-tree + GPU ctx."""
    pts = synth.nodes(10_000, 3)
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    ctx = Context(3)
    first = ctx.nodes_append(pts)
    assert first == 0 and ctx.n_nodes == 10_000
    yield pts, tree, ctx
    ctx.close()


def _oracle_lists(tree, Q, r):
    out = []
    for i, q in enumerate(Q):
        ri = r if np.isscalar(r) else r[i]
        idx, key = tree.within_range(float(ri), q)
        o = np.argsort(idx, kind="stable")
        out.append((idx[o], key[o]))
    return out


def _check_csr(offsets, idx, dist, ref):
    assert offsets[0] == 0 and offsets[-1] == len(idx)
    for i, (ri, rk) in enumerate(ref):
        a, b = offsets[i], offsets[i + 1]
        assert np.array_equal(idx[a:b], ri), f"query {i}: neighbour set differs"
        assert np.array_equal(dist[a:b], rk), f"query {i}: stored keys differ (bit-exact required)"


def test_device_sqrt_div(hip_lib):
    """fp64 sqrt and divide on gfx950 must be correctly rounded (the parity of every
    stored distance and of t = dot/edgeLen rests on it).  simple_steer exposes sqrt."""
    rng = np.random.default_rng(7)
    with Context(3) as ctx:
        a = np.zeros((200_000, 3))
        b = np.zeros((200_000, 3))
        b[:, 0] = rng.uniform(0, 100, 200_000)
        b[:, 1] = rng.uniform(0, 1e-3, 200_000)
        b[:, 2] = rng.uniform(0, 1e6, 200_000)
        dist, _ = ctx.simple_steer(a, b)
        ref = np.sqrt((b[:, 0] * b[:, 0] + b[:, 1] * b[:, 1]) + b[:, 2] * b[:, 2])
        assert np.array_equal(dist, ref)


def test_radius_c2_bit_exact(small3, oracle):
    pts, tree, ctx = small3
    Q = synth.queries(1024, 3)
    r = synth.ball_radius(10_000, 3)
    assert abs(r - 7.783652738915254) < 1e-12
    offsets, idx, dist = ctx.nn_radius(Q, r)
    _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))
    assert len(idx) > 1024  # non-trivial lists


def test_radius_long_lists_every_length(oracle):
    """The long-list build of the finishing kernel places a list's entries by counting inside index bins
    (no sort network): lists of 65 ... 8192 entries and the rank-counting path beyond, with indices that are
    spread evenly (uniform background), packed into a few bins (a cluster appended as one block next to a lone
    far index), and every entry in ONE bin (a cluster whose indices are consecutive)."""
    rng = np.random.default_rng(77)
    bg = rng.uniform(-400, 400, (150_000, 3))
    cl = rng.normal(0, 1.0, (12_000, 3)) + [900.0, 900.0, 900.0]       # indices 150000..161999, far from the rest
    lone = np.array([[900.0, 900.0, 905.0]])
    pts = np.r_[lone, bg, cl]                                            # index 0 sits inside the cluster's ball
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    c = np.array([900.0, 900.0, 900.0])
    Q = np.r_[np.tile(c, (10, 1)) + rng.normal(0, 0.2, (10, 3)), rng.uniform(-400, 400, (6, 3))]
    r = np.array([0.35, 0.5, 0.7, 1.0, 1.4, 2.0, 2.6, 3.2, 4.9, 30.0, 40.0, 60.0, 90.0, 120.0, 150.0, 5.0])
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        offsets, idx, dist = ctx.nn_radius(Q, r)
        lens = np.diff(offsets)
        assert lens.max() > 8192 and np.any((lens > 64) & (lens <= 2048)) and np.any((lens > 2048) & (lens <= 8192))
        _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))


def test_radius_long_lists_crowded_index_bins(oracle):
    """A block of consecutive indices next to a lone far index in a large tree crowds the 2048 index bins of the
    counting placement (hundreds of entries in one bin): such a list goes to the sort network instead; lists of the
    same call that do spread out are still placed by counting.  Against the oracle."""
    rng = np.random.default_rng(78)
    bg = rng.uniform(-400, 400, (594_000, 3))
    cl = rng.normal(0, 1.0, (6_000, 3)) + [900.0, 900.0, 900.0]        # indices 594001..600000
    lone = np.array([[900.0, 900.0, 903.0]])
    pts = np.r_[lone, bg, cl]
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    c = np.array([900.0, 900.0, 900.0])
    Q = np.r_[np.tile(c, (4, 1)) + rng.normal(0, 0.2, (4, 3)), rng.uniform(-400, 400, (2, 3))]
    r = np.array([1.0, 2.0, 3.2, 30.0, 60.0, 80.0])
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        offsets, idx, dist = ctx.nn_radius(Q, r)
        lens = np.diff(offsets)
        assert np.sum((lens > 512) & (lens <= 8192)) >= 4
        _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))


def test_radius_capacity_two_call(small3):
    pts, tree, ctx = small3
    Q = synth.queries(64, 3)
    offsets, idx, dist = ctx.nn_radius(Q, 7.78, cap=8)   # forces RRTX_E_CAPACITY then retry
    assert offsets[-1] == len(idx) > 8


def test_extend_candidates_capacity_two_call(small3, oracle):
    """a too-small capacity must come back as RRTX_E_CAPACITY without touching memory through the
    partly written lists (regression: the candidate kernel once indexed nodes with unwritten entries)"""
    pts, tree, ctx = small3
    ctx.spheres_set(synth.spheres(16))
    Q = synth.queries(128, 3)
    out = ctx.extend_candidates(Q, 7.78, ROBOT_RADIUS, cap=8)
    ref = ctx.extend_candidates(Q, 7.78, ROBOT_RADIUS, cap=1 << 16)
    assert len(out["idx"]) == len(ref["idx"]) > 8
    for k in ref:
        assert np.array_equal(out[k], ref[k]), k


def test_radius_per_query_radii_and_edge_cases(small3, oracle):
    pts, tree, ctx = small3
    rng = np.random.default_rng(3)
    Q = synth.queries(200, 3, seed=99)
    r = rng.uniform(0.0, 12.0, 200)
    r[0] = 0.0          # empty
    r[1] = 1e-300
    r[2] = 200.0        # everything
    Q[3] = pts[17]      # query on a node: dist 0 < r
    r[3] = 1e-9
    offsets, idx, dist = ctx.nn_radius(Q, r)
    _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))
    assert offsets[1] - offsets[0] == 0
    assert offsets[3] - offsets[2] == 10_000


@pytest.mark.parametrize("scale", [1.0, 1e3, 1e6])
def test_radius_prefilter_is_conservative(oracle, scale):
    """fp32 prefilter + exact confirm == exact scan == oracle, on inputs built to sit on the range
    boundary (|dist - r| ~ ulps) and on coordinates far from the origin (fp32 cancellation)."""
    from rrtqx_3d_amd import _capi
    rng = np.random.default_rng(123)
    nq, per = 32, 400
    r = 3.1497206024977595
    Q = rng.uniform(-50, 50, (nq, 3)) * scale
    pts = [np.zeros((1, 3))]
    for q in Q:
        u = rng.normal(size=(per, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        # distances straddling r by 0, +-1e-15 ... +-1e-9 relative, plus clear in/out points
        rel = rng.choice([0.0, 1e-16, -1e-16, 1e-15, -1e-15, 1e-13, -1e-13, 1e-9, -1e-9, 0.3, -0.3], per)
        pts.append(q + u * (r * (1.0 + rel))[:, None])
    pts = np.concatenate(pts, 0)
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    ref = _oracle_lists(tree, Q, r)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        for flt in (1, 0):
            ctx.set_option(_capi.RRTX_OPT_NN_FILTER, flt)
            offsets, idx, dist = ctx.nn_radius(Q, r)
            _check_csr(offsets, idx, dist, ref)
    if scale == 1.0:
        on_edge = sum(int(np.sum(np.abs(k - r) < 1e-12)) for _, k in ref)
        assert on_edge > 100     # the boundary cases really are exercised


def test_radius_filter_off_matches(small3, oracle):
    from rrtqx_3d_amd import _capi
    pts, tree, ctx = small3
    Q = synth.queries(256, 3)
    r = synth.ball_radius(10_000, 3)
    ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
    try:
        offsets, idx, dist = ctx.nn_radius(Q, r)
    finally:
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 1)
    _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))


def test_radius_nonfinite_coordinates(oracle):
    """inf / NaN / 1e300 coordinates switch the screen off (thr_f = inf); results still exact"""
    pts = np.array([[0, 0, 0], [1, 0, 0], [1e300, 0, 0], [1e300, 1, 0], [np.inf, 0, 0], [np.nan, 0, 0]])
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    Q = np.array([[0.5, 0, 0], [1e300, 0.5, 0]])
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        offsets, idx, dist = ctx.nn_radius(Q, 2.0)
        _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, 2.0))
        assert list(idx) == [0, 1, 2, 3]


def test_radius_root_inclusive_rule(oracle):
    """K2: the root is included with <=, every other node with <."""
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0], [3, 4, 0]])
        off, idx, dist = ctx.nn_radius([[0, 0, 0], [3, 4, 0]], 5.0)
        assert list(idx[off[0]:off[1]]) == [0]
        assert list(idx[off[1]:off[2]]) == [0, 1] and list(dist[off[1]:off[2]]) == [5.0, 0.0]


def test_radius_single_query_large_ball(small3, oracle):
    """obstacle sweep style query: one centre, r = robotRadius + delta + ob.radius (R/DRRT_Q.jl:3203)"""
    pts, tree, ctx = small3
    q = np.array([[1.0, -2.0, 3.0]])
    r = 0.5 + 8.0 + 3.5
    offsets, idx, dist = ctx.nn_radius(q, r)
    _check_csr(offsets, idx, dist, _oracle_lists(tree, q, r))
    assert len(idx) > 64   # exercises the k > 64 ordering path


def test_nearest_c2(small3):
    pts, tree, ctx = small3
    Q = synth.queries(1024, 3)
    idx, dist = ctx.nn_nearest(Q)
    for i, q in enumerate(Q):
        ri, rd = tree.nearest(q)
        assert idx[i] == ri and dist[i] == rd


def test_nearest_screened_vs_exact_and_adversarial_order(oracle):
    """the screened nearest scan == the exact fp64 scan == the oracle, including (a) exact ties in
    distance (lowest index wins), (b) a visiting order that makes every node a new running minimum
    (nodes sorted by decreasing distance: candidate overflow -> automatic exact fallback),
    (c) coordinates far from the origin."""
    from rrtqx_3d_amd import _capi
    rng = np.random.default_rng(21)
    for scale, n in ((1.0, 20_000), (1e4, 5_000)):
        pts = rng.uniform(-50, 50, (n, 3)) * scale
        pts[100] = pts[7]                                   # exact duplicate: tie on d2, index 7 must win
        Q = np.concatenate([rng.uniform(-50, 50, (500, 3)) * scale, pts[7:8] + 1e-3 * scale, pts[200:201]])
        tree = oracle.KDTree(3)
        tree.insert_many(pts)
        with Context(3) as ctx:
            ctx.nodes_append(pts)
            i1, d1 = ctx.nn_nearest(Q)
            ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
            i0, d0 = ctx.nn_nearest(Q)
            assert np.array_equal(i1, i0) and np.array_equal(d1, d0)
            for k in range(0, len(Q), 7):
                ri, rd = tree.nearest(Q[k], naive=True)
                assert d1[k] == rd and (i1[k] == ri or np.array_equal(pts[i1[k]], pts[ri]))
            assert i1[500] == 7 and i1[501] == 200 and d1[501] == 0.0
    # adversarial order: distances to the query strictly decreasing with the index
    n = 60_000
    q = np.zeros((4, 3))
    radii = np.linspace(90.0, 1.0, n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pts = u * radii[:, None]
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        idx, dist = ctx.nn_nearest(q)                       # overflows the candidate buffer -> exact scan
        ref = np.sqrt((pts[:, 0] * pts[:, 0] + pts[:, 1] * pts[:, 1]) + pts[:, 2] * pts[:, 2])
        assert (idx == np.argmin(ref)).all() and (dist == ref.min()).all()


def test_edges_spheres_bit_exact(small3, oracle):
    pts, tree, ctx = small3
    sph = synth.spheres(32)
    active = np.ones(32, dtype=np.uint8)
    active[5] = 0
    ctx.spheres_set(sph, active)
    osph, m = oracle.make_spheres(sph, active)
    Q = synth.queries(256, 3)
    r = synth.ball_radius(10_000, 3)
    offsets, idx, _ = ctx.nn_radius(Q, r)
    p0, p1 = synth.candidate_edges(Q, pts, offsets, idx)
    hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS)
    rh, rf = oracle.edges_check_spheres(osph, m, p0, p1, ROBOT_RADIUS)
    assert np.array_equal(hit, rh)
    assert np.array_equal(first, rf)
    assert 0 < hit.sum() < len(hit)
    # direction dependence of the reference's dot/edgeLen formula shows up
    n = len(idx)
    assert (hit[:n] != hit[n:]).any()
    # single-obstacle form (explicitEdgeCheck(S, edge, ob))
    h7, _ = ctx.edges_check(p0, p1, ROBOT_RADIUS, obstacle=7)
    one, m1 = oracle.make_spheres(sph[7:8])
    r7, _ = oracle.edges_check_spheres(one, m1, p0, p1, ROBOT_RADIUS)
    assert np.array_equal(h7, r7)
    h5, _ = ctx.edges_check(p0, p1, ROBOT_RADIUS, obstacle=5)   # inactive obstacle
    assert h5.sum() == 0


def test_edges_spheres_kats(oracle):
    """K3-K6 of SURVEY.md 8(c) through the GPU path."""
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.spheres_set([[0, 0, 0, 1.0]])
        p0 = np.array([[-2, 1.4, 0], [0.9, 0, 0], [10, 10, 10]], dtype=np.float64)
        p1 = np.array([[2, 1.4, 0], [1.4, 0, 0], [10, 10, 10]], dtype=np.float64)
        hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS)
        assert list(hit) == [0, 1, 1]       # K3 quirk no-hit, K4 hit, K5 zero-length => NaN => hit
        assert list(first) == [-1, 0, 0]
        ctx.spheres_set([[0, 0, 0, 1.0]], active=[0])
        hit, _ = ctx.edges_check(p0, p1, ROBOT_RADIUS)
        assert hit.sum() == 0               # K6


def test_obstacle_update(small3, oracle):
    pts, tree, ctx = small3
    sph = synth.spheres(32)
    ctx.spheres_set(sph)
    Q = synth.queries(64, 3)
    offsets, idx, _ = ctx.nn_radius(Q, 7.78)
    p0, p1 = synth.candidate_edges(Q, pts, offsets, idx)
    ctx.obstacle_update(3, sph[3, 3] + 2.0, True)       # obstacleAugmentation
    ctx.obstacle_update(4, sph[4, 3], False)            # expiry
    sph2 = sph.copy()
    sph2[3, 3] += 2.0
    act = np.ones(32, dtype=np.uint8)
    act[4] = 0
    osph, m = oracle.make_spheres(sph2, act)
    hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS)
    rh, rf = oracle.edges_check_spheres(osph, m, p0, p1, ROBOT_RADIUS)
    assert np.array_equal(hit, rh) and np.array_equal(first, rf)


def test_points_spheres(small3, oracle):
    pts, tree, ctx = small3
    sph = synth.spheres(64)
    ctx.spheres_set(sph)
    osph, m = oracle.make_spheres(sph)
    P = synth.queries(4096, 3, seed=5)
    for quick in (True, False):
        unsafe, clr = ctx.points_check(P, ROBOT_RADIUS, quick=quick)
        ru, rc = oracle.points_check_spheres(osph, m, P, ROBOT_RADIUS, quick=quick)
        assert np.array_equal(unsafe, ru)
        assert np.array_equal(clr, rc)
    assert 0 < unsafe.sum() < len(unsafe)


def test_extend_candidates_matches_parts(small3, oracle):
    pts, tree, ctx = small3
    sph = synth.spheres(32)
    ctx.spheres_set(sph)
    osph, m = oracle.make_spheres(sph)
    Q = synth.queries(512, 3)
    r = synth.ball_radius(10_000, 3)
    out = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
    ref = _oracle_lists(tree, Q, r)
    _check_csr(out["offsets"], out["idx"], out["cost"], ref)
    p0, p1 = synth.candidate_edges(Q, pts, out["offsets"], out["idx"])
    rh, _ = oracle.edges_check_spheres(osph, m, p0, p1, ROBOT_RADIUS)
    n = len(out["idx"])
    assert np.array_equal(out["hit_out"], rh[:n])
    assert np.array_equal(out["hit_in"], rh[n:])
    for i, q in enumerate(Q):
        ri, rd = tree.nearest(q)
        assert out["nearest_idx"][i] == ri and out["nearest_dist"][i] == rd
    ru, _ = oracle.points_check_spheres(osph, m, Q, ROBOT_RADIUS, quick=True)
    assert np.array_equal(out["sample_unsafe"], ru)


@pytest.mark.parametrize("root_at_cluster", [True, False])
def test_candidate_edges_far_from_origin(oracle, root_at_cluster):
    """the packed fp32 reach screen of the fused kernel works on coordinates relative to the first
    node; put the whole scene 1e5 away (and, in the second case, leave the root at 0 so the shifted
    coordinates are large and fp32 cancellation is at its worst) and require bit-exact flags"""
    rng = np.random.default_rng(31)
    off = np.array([1.0e5, -2.0e5, 3.0e5])
    pts = rng.uniform(-20, 20, (4000, 3)) + off
    if not root_at_cluster:
        pts[0] = 0.0
    sph = np.concatenate([rng.uniform(-20, 20, (48, 3)) + off, rng.uniform(1.0, 3.5, (48, 1))], 1)
    Q = rng.uniform(-20, 20, (256, 3)) + off
    Q[:8] = sph[:8, :3] + rng.normal(0, 1.0, (8, 3))          # samples right at obstacles
    osph, m = oracle.make_spheres(sph)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph)
        out = ctx.extend_candidates(Q, 8.0, ROBOT_RADIUS)
        p0, p1 = synth.candidate_edges(Q, pts, out["offsets"], out["idx"])
        rh, _ = oracle.edges_check_spheres(osph, m, p0, p1, ROBOT_RADIUS)
        n = len(out["idx"])
        assert n > 1000 and 0 < rh.sum() < 2 * n
        assert np.array_equal(out["hit_out"], rh[:n]) and np.array_equal(out["hit_in"], rh[n:])


@pytest.mark.parametrize("case", ["dense", "huge_radius", "on_nodes", "inactive_mix"])
def test_extend_candidates_sphere_lists(oracle, case):
    """the candidate edges only look at the spheres on their sample's short list; lists that
    overflow (dense or huge spheres, huge search radius), zero-length edges (a sample sitting on a
    node collides with every active sphere, R/DRRT_Q.jl:1208) and inactive spheres must give the
    reference's flags all the same"""
    rng = np.random.default_rng({"dense": 1, "huge_radius": 2, "on_nodes": 3, "inactive_mix": 4}[case])
    pts = rng.uniform(-20, 20, (6000, 3))
    Q = rng.uniform(-20, 20, (300, 3))
    r = 4.0
    active = None
    if case == "dense":
        sph = np.concatenate([rng.uniform(-20, 20, (1500, 3)), rng.uniform(0.5, 6.0, (1500, 1))], 1)
    elif case == "huge_radius":
        sph = np.concatenate([rng.uniform(-20, 20, (40, 3)), rng.uniform(0.5, 3.0, (40, 1))], 1)
        sph[3, 3] = 60.0          # one sphere swallows the world
        r = 45.0
        Q = Q[:24]
    elif case == "on_nodes":
        sph = np.concatenate([rng.uniform(-20, 20, (30, 3)), rng.uniform(0.5, 3.0, (30, 1))], 1)
        Q[:100] = pts[rng.integers(0, len(pts), 100)]          # zero-length candidate edges
        Q[100:110] = pts[0]                                    # ... to the root as well
    else:
        sph = np.concatenate([rng.uniform(-20, 20, (200, 3)), rng.uniform(0.5, 5.0, (200, 1))], 1)
        active = (rng.uniform(size=200) < 0.5).astype(np.uint8)
    osph, m = oracle.make_spheres(sph, active) if active is not None else oracle.make_spheres(sph)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph, active) if active is not None else ctx.spheres_set(sph)
        out = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
        p0, p1 = synth.candidate_edges(Q, pts, out["offsets"], out["idx"])
        rh, _ = oracle.edges_check_spheres(osph, m, p0, p1, ROBOT_RADIUS)
        n = len(out["idx"])
        assert n > 500
        assert np.array_equal(out["hit_out"], rh[:n]) and np.array_equal(out["hit_in"], rh[n:])
        ru, _ = oracle.points_check_spheres(osph, m, Q, ROBOT_RADIUS, quick=True)
        assert np.array_equal(out["sample_unsafe"], ru)
        # the stand-alone edge kernel (full obstacle loop) agrees too
        h2, _ = ctx.edges_check(p0, p1, ROBOT_RADIUS)
        assert np.array_equal(h2, rh)
        if case == "on_nodes":
            assert rh[:n][out["cost"] == 0.0].all()            # every zero-length edge "collides"


def test_polygons_edges_and_points(oracle):
    polys = synth.polygons(64)
    ps = oracle.PolygonSet(polys)
    with Context(4) as ctx:
        ctx.nodes_append(synth.nodes(16, 4))
        ctx.polygons_set(polys)
        rng = np.random.default_rng(11)
        p0 = synth.nodes(4096, 4, seed=21)
        step = rng.normal(0, 6.0, size=(4096, 2))
        p1 = p0.copy()
        p1[:, :2] += step
        p1[:64, 0] = p0[:64, 0]          # near-vertical edges exercise the abs(dx) < 1e-6 branch
        hit, first = ctx.edges_check(p0, p1, ROBOT_RADIUS, kind=1)
        rh, rf = oracle.edges_check_polygons(ps, p0, p1, ROBOT_RADIUS)
        assert np.array_equal(hit, rh) and np.array_equal(first, rf)
        assert 0 < hit.sum() < len(hit)
        unsafe, clr = ctx.points_check(p0, ROBOT_RADIUS, kind=1)
        ru = np.zeros(len(p0), dtype=np.uint8)
        rc = np.zeros(len(p0))
        for i in range(len(p0)):
            u, c = oracle.point_check_polygons(ps, p0[i], ROBOT_RADIUS)
            ru[i], rc[i] = u, c
        assert np.array_equal(unsafe, ru)
        assert np.array_equal(clr, rc)


@pytest.mark.parametrize("moving", [True, False])
def test_polygon_point_flag_without_certificate_matches_full_check(oracle, moving):
    """explicitPointCheck with no certificate wanted (clearance == NULL) must return the flag of the call that
    wants it -- random points, points at the robot radius from a polygon edge or a ball to within ulps, points on
    bounding circles, moving obstacles, non-finite points.  (Until the end of round 2 the flag-only call walked the
    near obstacles only; the lattice scenes showed that wrong, tests/test_gpu_lattice.py, K12.)  Without moving
    obstacles the flag-only call is a kernel of its own (two points per wave, the reference's loop as its fall-back:
    an odd number of points, points at the height of a vertex and points inside far bounding boxes go through it)."""
    rng = np.random.default_rng(78)
    polys, kinds, paths = [], [], []
    for i in range(200):
        c = rng.uniform(-40, 40, 2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, rng.integers(3, 7)))
        polys.append(c + np.c_[np.cos(ang), np.sin(ang)] * rng.uniform(0.5, 6))
        k = 1 if i % 5 == 0 else (6 if i % 11 == 3 and moving else 3)
        kinds.append(k)
        paths.append(np.c_[rng.uniform(-10, 10, (4, 2)), np.sort(rng.uniform(0, 30, 4))] if k == 6 else None)
    ps = oracle.PolygonSet(polys, kinds=kinds, paths=paths)
    cr = ps.centre_radius()
    n = 30_000 if moving else 30_001
    P = np.zeros((n, 3))
    P[:, :2] = rng.uniform(-45, 45, (n, 2))
    P[:, 2] = rng.uniform(0, 30, n)
    rr = 0.5
    for i in range(6000):                       # at distance rr (1 +- ulps) from an edge of a static polygon
        j = i % 200
        if kinds[j] != 3:
            continue
        v = polys[j]
        a, b = v[i % len(v)], v[(i + 1) % len(v)]
        t = rng.uniform(0.1, 0.9)
        e = b - a
        nrm = np.array([e[1], -e[0]]) / np.hypot(*e)
        P[i, :2] = a + t * e + nrm * rr * (1.0 + (i % 9 - 4) * 2.0 ** -51) * (1 if i % 2 else -1)
    for i in range(6000, 9000):                 # on the bounding circle inflated by the robot radius, +- ulps
        j = i % 200
        t = rng.uniform(0, 2 * np.pi)
        P[i, :2] = cr[j, :2] + np.array([np.cos(t), np.sin(t)]) * (cr[j, 2] + rr) * (1.0 + (i % 7 - 3) * 2.0 ** -52)
    P[9000] = [np.nan, 0, 0]; P[9001] = [np.inf, 1, 2]; P[9002] = [1e300, -1e300, 5]
    for i in range(9003, 11000):                # at the height of a vertex of some polygon, anywhere along x
        P[i, 1] = polys[i % 200][i % len(polys[i % 200]), 1]
    P[12000:12040, :2] *= 3.0                   # outside the grid over the obstacles (the whole list is walked), ...
    P[12100:12108, 0] += 200.0                  # ... whole workgroups of them and single ones among points inside
    with Context(3) as ctx:
        ctx.nodes_append(synth.nodes(16, 3))
        ctx.polygons_set(polys, kinds=kinds, paths=paths if moving else None)
        full, clr = ctx.points_check(P, rr, kind=1)
        flag, none = ctx.points_check(P, rr, kind=1, want_clearance=False)
        assert none is None
        assert np.array_equal(full, flag)
        assert 0 < full.sum() < n
        for i in list(range(0, 11000, 37)) + [9000, 9001, 9002, n - 1] + list(range(12000, 12040, 3)) + [12100, 12107]:       # and the full check is the oracle's
            u, c = oracle.point_check_polygons(ps, P[i], rr)
            assert full[i] == u and (clr[i] == c or (np.isnan(clr[i]) and np.isnan(c)))


def test_polygons_edges_degenerate_and_boundary_inputs(oracle):
    """The polygon edge kernel drops far obstacles with a box test before the reference's tests run;
    that shortcut must never change an answer: NaN / inf / huge coordinates, zero-length edges, edges
    that graze a bounding circle within rounding, balls (kind 1), many obstacles (several groups of
    32), negative robot radius -- all against the oracle, first-hit indices included."""
    rng = np.random.default_rng(77)
    polys, kinds = [], []
    for i in range(150):
        c = rng.uniform(-40, 40, 2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, rng.integers(3, 7)))
        polys.append(c + np.c_[np.cos(ang), np.sin(ang)] * rng.uniform(0.5, 6))
        kinds.append(1 if i % 5 == 0 else 3)
    ps = oracle.PolygonSet(polys, kinds=kinds)
    cr = ps.centre_radius()
    n = 6000
    p0 = np.zeros((n, 3)); p1 = np.zeros((n, 3))
    p0[:, :2] = rng.uniform(-45, 45, (n, 2))
    p1[:, :2] = p0[:, :2] + rng.normal(0, 5.0, (n, 2))
    # grazing edges: tangent to a bounding circle at distance (robot + radius) * (1 +- a few ulp)
    for i in range(1500):
        j = i % len(polys)
        for_r = 0.5 + cr[j, 2]
        d = for_r * (1.0 + (i % 7 - 3) * 2.0 ** -52)
        t = rng.uniform(0, 2 * np.pi)
        nrm = np.array([np.cos(t), np.sin(t)]); tan = np.array([-nrm[1], nrm[0]])
        mid = cr[j, :2] + nrm * d
        p0[i, :2] = mid - tan * 2.0; p1[i, :2] = mid + tan * 2.0
    # edges parallel to a polygon side at distance robot * (1 +- a few ulp) (segment-level threshold), and
    # edges exactly collinear with a side of an axis-aligned box but disjoint from it: the reference's
    # side tests then say "crossing" and segmentDistSqrd returns 0.0 (R/DRRT.jl:1149-1193)
    boxes = [np.array([[x, y], [x + 4, y], [x + 4, y + 2], [x, y + 2]], dtype=np.float64)
             for x, y in ((-30.0, -30.0), (10.0, 20.0), (25.0, -12.0))]
    polys += boxes; kinds += [3, 3, 3]
    ps = oracle.PolygonSet(polys, kinds=kinds)
    k = 1600
    for bx in boxes:
        x0, y0 = bx[0]
        for i in range(7):
            off = 0.5 * (1.0 + (i - 3) * 2.0 ** -52)
            p0[k, :2] = [x0 + 0.5, y0 - off]; p1[k, :2] = [x0 + 3.5, y0 - off]; k += 1    # below the bottom side
            p0[k, :2] = [x0 + 4 + off, y0 + 0.25]; p1[k, :2] = [x0 + 4 + off, y0 + 1.75]; k += 1   # right of the right side
        p0[k, :2] = [x0 + 4.75, y0]; p1[k, :2] = [x0 + 5.5, y0]; k += 1               # collinear with the bottom side
        p0[k, :2] = [x0 - 1.5, y0 + 2]; p1[k, :2] = [x0 - 0.75, y0 + 2]; k += 1       # collinear with the top side
        p0[k, :2] = [x0, y0 - 1.5]; p1[k, :2] = [x0, y0 - 0.75]; k += 1               # collinear with the left side
        # the same inside the bounding circle (robot radius 0.5) but more than the robot radius away from the side:
        # nothing but the side tests themselves tells the reference's answer (0.0, a hit, for the horizontal ones;
        # the "close to vertical" branch is not strict and separates the vertical one)
        p0[k, :2] = [x0 + 4.51, y0]; p1[k, :2] = [x0 + 4.515, y0]; k += 1
        p0[k, :2] = [x0 - 0.515, y0 + 2]; p1[k, :2] = [x0 - 0.51, y0 + 2]; k += 1
        p0[k, :2] = [x0, y0 - 0.515]; p1[k, :2] = [x0, y0 - 0.51]; k += 1
    # ... and on the extension of a slanted side, where the slope and the differences round: every variant one ulp
    # off in one coordinate must still come out as the reference's own operations say
    tri = np.array([[50.0, 50.0], [53.0, 54.0], [49.0, 57.0]])
    polys.append(tri); kinds.append(3)
    ps = oracle.PolygonSet(polys, kinds=kinds)
    for i in range(240):
        a = np.array([53.75, 55.0]); b = np.array([54.125, 55.5])
        if i >= 120:                                                 # beyond the other end of the side
            a = np.array([49.578125, 49.4375]); b = np.array([49.53125, 49.375])
        v = i % 120
        if v > 0:
            tgt = (a, b)[(v >> 1) & 1]
            c = (v >> 2) & 1
            for _ in range(1 + (v >> 3)):
                tgt[c] = np.nextafter(tgt[c], np.inf if v & 1 else -np.inf)
        p0[k, :2] = a; p1[k, :2] = b; k += 1
    p1[1500:1540] = p0[1500:1540]                                   # zero-length edges
    p0[1540:1550, 0] = np.nan; p1[1550:1560, 1] = np.nan            # NaN endpoints
    p0[1560:1570, 0] = np.inf; p1[1570:1580, 1] = -np.inf
    p0[1580:1590, :2] = 1e300; p1[1590:1600, 0] = -1e300
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.polygons_set(polys, kinds=kinds)
        for rr in (0.5, 0.0, 3.0, -2.0):
            hit, first = ctx.edges_check(p0, p1, rr, kind=1)
            rh, rf = oracle.edges_check_polygons(ps, p0, p1, rr)
            assert np.array_equal(hit, rh) and np.array_equal(first, rf)
            if rr == 0.5:
                # the far collinear edges do "cross" in the reference, the vertical one does not
                assert rh[k - 240] == 1 and rh[k - 120] == 1 and list(rh[k - 243:k - 240]) == [1, 1, 0]
        assert 0 < hit.sum() < len(hit)


@pytest.mark.parametrize("with_moving", [False, True])
def test_extend_candidates_against_polygon_list(oracle, with_moving):
    """RRTX_OPT_EXTEND_OBSTACLES = 1: the fused preamble checks both directed candidate edges and the
    samples against the polygon list (the candidate edges come straight from the CSR lists on the
    device); compared with the one-at-a-time oracle calls.  With moving polygons the third coordinate
    of samples and nodes is time."""
    rng = np.random.default_rng(31)
    n, b = 20000, 700
    pts = rng.uniform(-50, 50, (n, 3))
    Q = rng.uniform(-50, 50, (b, 3))
    polys = synth.polygons(48)
    kinds, paths, active = [3] * 48, None, [1] * 48
    active[5] = 0
    if with_moving:
        pts[:, 2] = rng.uniform(0, 40, n); Q[:, 2] = rng.uniform(0, 40, b)
        kinds = [6 if i % 4 == 1 else 3 for i in range(48)]
        paths = [None if k == 3 else np.c_[rng.uniform(-20, 20, (5, 2)), np.sort(rng.uniform(0, 40, 5))] for k in kinds]
    ps = oracle.PolygonSet(polys, kinds=kinds, active=active, paths=paths)
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    r = 6.0
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, kinds=kinds, active=active, paths=paths)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        out = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
        off, idx = out["offsets"], out["idx"]
        _check_csr(off, idx, out["cost"], _oracle_lists(tree, Q, r))
        p0, p1 = synth.candidate_edges(Q, pts, off, idx)
        rh, _ = oracle.edges_check_polygons(ps, p0, p1, ROBOT_RADIUS)
        k = len(idx)
        assert k > 5000 and np.array_equal(out["hit_out"], rh[:k]) and np.array_equal(out["hit_in"], rh[k:])
        assert 0 < rh.sum() < len(rh)
        exp_unsafe = np.array([oracle.point_check_polygons(ps, qq, ROBOT_RADIUS)[0] for qq in Q])
        assert np.array_equal(out["sample_unsafe"].astype(bool), exp_unsafe)
        for i in range(b):
            if off[i + 1] > off[i]:
                ni, nd = tree.nearest(Q[i])
                assert out["nearest_idx"][i] == ni and out["nearest_dist"][i] == nd
        # same context, back to the sphere list (empty): nothing collides
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 0)
        out0 = ctx.extend_candidates(Q, r, ROBOT_RADIUS)
        assert not out0["hit_out"].any() and not out0["sample_unsafe"].any()


@pytest.mark.parametrize("seed", range(4))
def test_extend_polygon_path_equals_explicit_edge_and_point_checks(seed):
    """The fused preamble's polygon checks take shortcuts the stand-alone entry points do not (a wave of candidate
    edges walks only the obstacles near its samples; the sample check is a flag-only call): on random
    scenes -- many obstacles, balls, moving and inactive ones, short and long radii, samples on top of nodes
    (zero-length edges), a NaN sample -- both must give what rrtx_edges_check / rrtx_points_check give for the
    same edges and points (those walk the whole list and are held to the oracle elsewhere)."""
    rng = np.random.default_rng(900 + seed)
    m = int(rng.integers(40, 400))
    polys, kinds, paths, active = [], [], [], []
    for i in range(m):
        c = rng.uniform(-50, 50, 2)
        ang = np.sort(rng.uniform(0, 2 * np.pi, rng.integers(3, 7)))
        polys.append(c + np.c_[np.cos(ang), np.sin(ang)] * rng.uniform(0.3, 9))
        k = 1 if i % 7 == 0 else (6 if (seed % 2 == 1 and i % 9 == 4) else 3)
        kinds.append(k)
        paths.append(np.c_[rng.uniform(-15, 15, (4, 2)), np.sort(rng.uniform(0, 30, 4))] if k == 6 else None)
        active.append(0 if i % 13 == 5 else 1)
    n, b = 30_000, 2500
    pts = rng.uniform(-50, 50, (n, 3)); pts[:, 2] = rng.uniform(0, 30, n)
    Q = rng.uniform(-50, 50, (b, 3)); Q[:, 2] = rng.uniform(0, 30, b)
    Q[:40] = pts[rng.integers(0, n, 40)]                 # samples on nodes
    Q[40] = [np.nan, 1.0, 2.0]
    r = float(rng.choice([1.5, 4.0, 9.0]))
    rr = float(rng.choice([0.0, 0.5, 2.0]))
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, kinds=kinds, active=active, paths=paths)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        out = ctx.extend_candidates(Q, r, rr)
        off, idx = out["offsets"], out["idx"]
        k = len(idx)
        assert k > b
        p0, p1 = synth.candidate_edges(Q, pts, off, idx)
        hit, _ = ctx.edges_check(p0, p1, rr, kind=1)
        assert np.array_equal(out["hit_out"], hit[:k]) and np.array_equal(out["hit_in"], hit[k:])
        unsafe, _ = ctx.points_check(Q, rr, kind=1)
        assert np.array_equal(out["sample_unsafe"], unsafe)
        assert 0 < hit.sum() < len(hit)


def test_polygon_kat_k7(oracle):
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.polygons_set([[[0, 0], [1, 0], [1, 1], [0, 1]]])
        hit, _ = ctx.edges_check([[-1, .5, 0], [-1, 2, 0]], [[2, .5, 0], [2, 2, 0]], 0.1, kind=1)
        assert list(hit) == [1, 0]


def test_polygon_kats_k11_k12(oracle):
    """The two reference quirks of tests/test_oracle_kat.py K11 / K12 through the C-ABI: an edge on the extension of
    a polygon side, inside the bounding circle and farther than the robot radius from the side, is a hit
    (segmentDistSqrd answers 0.0 for segments on one line); a point far left of a triangle at the height of its
    left vertex is "inside" (strict crossing tests) -- with and without the certificate, and also when an obstacle
    earlier in the list gives the loop a certificate first (the triangle is then skipped by the reference, and the
    point is safe)."""
    box = [[0.0, 0.0], [4.0, 0.0], [4.0, 2.0], [0.0, 2.0]]
    tri = [[0.0, 0.0], [2.0, -1.0], [2.0, 1.0]]
    near_box = [[-5.5, 2.0], [-4.5, 2.0], [-4.5, 3.0], [-5.5, 3.0]]   # 1.9 from the first point: certificate 1.9 < 4.49
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        ctx.polygons_set([box])
        p0 = [[4.51, 0.0, 0], [4.51, 0.001, 0], [0.0, -0.515, 0]]
        p1 = [[4.515, 0.0, 0], [4.515, 0.001, 0], [0.0, -0.51, 0]]
        hit, _ = ctx.edges_check(p0, p1, 0.5, kind=1)
        rh, _ = oracle.edges_check_polygons(oracle.PolygonSet([box]), np.array(p0), np.array(p1), 0.5)
        assert list(hit) == [1, 0, 0] and list(rh) == [1, 0, 0]
        pts = np.array([[-5.0, 0.0, 0], [-5.0, 0.5, 0], [5.0, 0.0, 0]])
        for polys, expect in (([tri], [1, 0, 0]), ([near_box, tri], [0, 0, 0]), ([tri, near_box], [1, 0, 0])):
            ctx.polygons_set(polys)
            ps = oracle.PolygonSet(polys)
            exp = [oracle.point_check_polygons(ps, p, 0.1) for p in pts]
            assert [int(e[0]) for e in exp] == expect
            unsafe, clr = ctx.points_check(pts, 0.1, kind=1)
            assert list(unsafe) == expect and np.array_equal(clr, np.array([e[1] for e in exp]))
            flag, _ = ctx.points_check(pts, 0.1, kind=1, want_clearance=False)
            assert list(flag) == expect


def test_radius_wrapped_theta_c3(oracle):
    """Dubins space [x y 0 theta], theta wraps at 2*pi (R/DRRT.jl:3312): ghost rule + dedupe."""
    n = 20_000
    pts = synth.nodes(n, 4)
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        Q = synth.queries(256, 4)
        for r in (10.0, 2.5):     # r > pi: both copies can see the same node; r < pi: ghosts mostly skipped
            offsets, idx, dist = ctx.nn_radius(Q, r)
            _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))
        idx, dist = ctx.nn_nearest(Q)
        for i, q in enumerate(Q):
            ri, rd = tree.nearest(q)
            assert idx[i] == ri and dist[i] == rd


def test_radius_two_wrapped_dimensions(oracle):
    """ghost iterator with two wrapped dimensions (3 ghosts per query, R/ghostPoint.jl:60-111):
    binary visiting order, skip rule and first-discovery keys against the oracle's kd-tree"""
    rng = np.random.default_rng(17)
    n = 6000
    pts = rng.uniform(0.0, 1.0, (n, 3))
    pts[:, 1] *= 10.0                       # the middle dimension does not wrap
    tree = oracle.KDTree(3, wraps=[0, 2], wrap_points=[1.0, 1.0])
    tree.insert_many(pts)
    Q = pts[rng.integers(0, n, 128)] + rng.normal(0, 0.01, (128, 3))
    Q[:, [0, 2]] = np.clip(Q[:, [0, 2]], 0.0, 1.0)
    with Context(3) as ctx:
        ctx.set_wrap(0, 1.0)
        ctx.set_wrap(2, 1.0)
        ctx.nodes_append(pts)
        for r in (0.08, 0.35, 0.7):         # 0.7 > period/2: the same node is seen by several copies
            offsets, idx, dist = ctx.nn_radius(Q, r)
            _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))
        idx, dist = ctx.nn_nearest(Q)
        for i, q in enumerate(Q):
            ri, rd = tree.nearest(q)
            assert dist[i] == rd and (idx[i] == ri or np.array_equal(pts[idx[i]], pts[ri]))


def test_detmath_device_equals_host(oracle):
    """include/rrtx_detmath.h compiled by hipcc for gfx950 == the same header compiled by gcc into the oracle, bit
    for bit: random arguments over the ranges the Dubins code produces, huge / tiny / non-finite ones, exact
    multiples of pi/4, zeros of both signs."""
    rng = np.random.default_rng(77)
    n = 400_000
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, np.inf, -np.inf, np.nan, 1e-310, -1e-310, 1e300, -1e300, 5e-324,
                        math.pi, -math.pi, math.pi / 2, -math.pi / 2, math.pi / 4, 3 * math.pi / 4, 2 * math.pi, 1e6, 1e7, 1.6e6])
    ang = np.concatenate([rng.uniform(-20, 20, n), rng.normal(0, 1, n) * 10.0 ** rng.integers(-12, 7, n),
                          np.arange(-64, 65) * (math.pi / 4), special])
    yy = np.concatenate([rng.normal(0, 1, n) * 10.0 ** rng.integers(-8, 8, n), np.repeat(special, len(special))])
    xx = np.concatenate([rng.normal(0, 1, n) * 10.0 ** rng.integers(-8, 8, n), np.tile(special, len(special))])
    ac = np.concatenate([rng.uniform(-1, 1, n), 1.0 - 10.0 ** rng.uniform(-17, 0, n), special])
    with Context(4) as ctx:
        for op, x, y in ((oracle.DM_SIN, ang, None), (oracle.DM_COS, ang, None), (oracle.DM_ATAN2, xx, yy), (oracle.DM_ACOS, ac, None)):
            dev = ctx.detmath_eval(op, x, y)
            host = oracle.dm_eval(op, x, y)
            same = (dev.view(np.uint64) == host.view(np.uint64)) | (np.isnan(dev) & np.isnan(host))
            assert same.all(), (op, np.flatnonzero(~same)[:5], x[~same][:5], dev[~same][:5], host[~same][:5])


def test_dubins_steer_exact(oracle):
    rng = np.random.default_rng(5)
    ne = 4096
    s = synth.nodes(ne, 4, seed=31)
    g = s.copy()
    g[:, :2] += rng.normal(0, 4.0, size=(ne, 2))
    g[:, 3] = rng.uniform(0, 2 * math.pi, ne)
    with Context(4) as ctx:
        ctx.nodes_append(s[:4])
        for r_min in (1.0, 2.0):
            cost, word = ctx.dubins_steer(s, g, r_min)
            for i in range(ne):
                c, w, _ = oracle.dubins_steer(s[i], g[i], r_min, want_traj=False)
                assert cost[i] == c and w.encode() == word[i], (i, cost[i], c, w, word[i])


def test_dubins_degenerate_poses_exact(oracle):
    """Poses where the last bit of atan2 decides: goals exactly ahead of / behind / beside the start on lattice
    coordinates, headings multiples of pi/4, identical poses, goals on the start's turning circles.  There a turn of
    length exactly 0 must not come out as a full turn on one side only (R/DRRT_distance_functions.jl:62-80, `theta < 0`):
    costs and words equal bit for bit, polylines row for row."""
    hd = np.arange(8) * (math.pi / 4)
    s_l, g_l = [], []
    for ti in hd:
        for tg in hd:
            for dx in (-4.0, -2.0, -1.0, -0.5, 0.0, 0.25, 0.5, 1.0, 2.0, 3.0, 4.0, 8.0):
                for dy in (-4.0, -2.0, -1.0, 0.0, 0.5, 1.0, 2.0, 4.0):
                    s_l.append([1.0, -2.0, 0.0, ti]); g_l.append([1.0 + dx, -2.0 + dy, 0.0, tg])
    s, g = np.array(s_l), np.array(g_l)
    with Context(4) as ctx:
        ctx.nodes_append(s[:2])
        for r_min in (0.5, 1.0, 2.0):
            cost, word = ctx.dubins_steer(s, g, r_min)
            off, xy = ctx.dubins_trajectory(s, g, r_min)
            full_turns = 0
            for i in range(len(s)):
                c, w, traj = oracle.dubins_steer(s[i], g[i], r_min)
                assert (cost[i] == c or (np.isnan(c) and np.isnan(cost[i]))) and w.encode() == word[i], (i, s[i], g[i], cost[i], c)
                assert np.array_equal(xy[off[i]:off[i + 1]], traj, equal_nan=True), i
                full_turns += c >= 2 * math.pi * r_min
            assert full_turns > 0


def test_extend_candidates_dubins(oracle):
    """fused Dubins preamble == its parts (wrapped range search + per-edge steer/check) == oracle"""
    n = 8_000
    pts = synth.nodes(n, 4)
    Q = synth.queries(96, 4)
    polys = synth.polygons(32)
    ps = oracle.PolygonSet(polys)
    r, r_min = 6.0, 1.0
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys)
        out = ctx.extend_candidates_dubins(Q, r, ROBOT_RADIUS, r_min)
        off, idx = out["offsets"], out["idx"]
        _check_csr(off, idx, out["key"], _oracle_lists(tree, Q, r))
        owner = np.repeat(np.arange(len(Q)), np.diff(off))
        s, g = Q[owner], pts[idx]
        co, wo, ho, _ = ctx.dubins_edges_check(s, g, r_min, ROBOT_RADIUS)       # same device code, per edge
        ci, wi, hi, _ = ctx.dubins_edges_check(g, s, r_min, ROBOT_RADIUS)
        assert np.array_equal(out["cost_out"], co) and np.array_equal(out["cost_in"], ci)
        assert np.array_equal(out["word_out"], wo) and np.array_equal(out["word_in"], wi)
        assert np.array_equal(out["hit_out"], ho) and np.array_equal(out["hit_in"], hi)
        assert (co != ci).any()                                                   # Dubins edges are directed
        for k in range(0, len(idx), max(1, len(idx) // 300)):
            c, w, traj = oracle.dubins_steer(s[k], g[k], r_min)
            h, _ = oracle.dubins_edge_check_polygons(ps, s[k], g[k], traj, ROBOT_RADIUS, r_min)
            assert co[k] == c and wo[k] == w.encode() and bool(ho[k]) == h, k
            c, w, traj = oracle.dubins_steer(g[k], s[k], r_min)
            h, _ = oracle.dubins_edge_check_polygons(ps, g[k], s[k], traj, ROBOT_RADIUS, r_min)
            assert ci[k] == c and wi[k] == w.encode() and bool(hi[k]) == h, k
        for i in range(len(Q)):
            ni, nd = tree.nearest(Q[i])
            assert out["nearest_idx"][i] == ni and out["nearest_dist"][i] == nd
        for i in range(0, len(Q), 5):
            u, _ = oracle.point_check_polygons(ps, Q[i], ROBOT_RADIUS)
            assert bool(out["sample_unsafe"][i]) == u


def test_dubins_trajectory_polyline(oracle):
    """edge.trajectory (R/DRRT_DubinsEdge_functions.jl:506-701): the rows the reference's float ranges produce,
    equal to the oracle's bit for bit."""
    rng = np.random.default_rng(12)
    ne = 512
    s = synth.nodes(ne, 4, seed=51)
    g = s.copy()
    g[:, :2] += rng.normal(0, 4.0, size=(ne, 2))
    g[:, 3] = rng.uniform(0, 2 * math.pi, ne)
    with Context(4) as ctx:
        ctx.nodes_append(s[:2])
        off, xy = ctx.dubins_trajectory(s, g, 1.0)
        _, _, _, tl = ctx.dubins_edges_check(s, g, 1.0, ROBOT_RADIUS)
        assert np.array_equal(np.diff(off), tl) and off[-1] == len(xy)
        for i in range(ne):
            _, w, traj = oracle.dubins_steer(s[i], g[i], 1.0)
            assert np.array_equal(xy[off[i]:off[i + 1]], traj), i


def test_dubins_edges_check(oracle):
    polys = synth.polygons(64)
    ps = oracle.PolygonSet(polys)
    rng = np.random.default_rng(6)
    ne = 2048
    s = synth.nodes(ne, 4, seed=41)
    g = s.copy()
    g[:, :2] += rng.normal(0, 5.0, size=(ne, 2))
    g[:, 3] = rng.uniform(0, 2 * math.pi, ne)
    r_min = 1.0
    with Context(4) as ctx:
        ctx.nodes_append(s[:4])
        ctx.polygons_set(polys)
        cost, word, hit, tl = ctx.dubins_edges_check(s, g, r_min, ROBOT_RADIUS)
        for i in range(ne):
            c, w, traj = oracle.dubins_steer(s[i], g[i], r_min)
            h, _ = oracle.dubins_edge_check_polygons(ps, s[i], g[i], traj, ROBOT_RADIUS, r_min)
            assert cost[i] == c and word[i] == w.encode() and bool(hit[i]) == h and tl[i] == traj.shape[0], i
        assert 0 < hit.sum() < ne


def test_pack_hits_layout():
    """rrtx_pack_hits_dev (the payload of the multi-GPU bitmask all-reduce): bit e = hit_out[e],
    bit cap+e = hit_in[e], entries at or beyond n_valid read 0."""
    import torch
    from rrtqx_3d_amd import parallel
    cap, n_valid = 1000, 777
    rng = np.random.default_rng(4)
    ho = (rng.random(cap) < 0.4).astype(np.uint8)
    hi = (rng.random(cap) < 0.4).astype(np.uint8)
    dev = torch.device("cuda", 0)
    d_ho, d_hi = torch.from_numpy(ho).to(dev), torch.from_numpy(hi).to(dev)
    d_nv = torch.tensor([n_valid], dtype=torch.int64, device=dev)
    words = torch.zeros(parallel.words_for(cap), dtype=torch.int64, device=dev)
    with Context(3) as ctx:
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        ctx.pack_hits_dev(d_ho.data_ptr(), d_hi.data_ptr(), d_nv.data_ptr(), cap, words.data_ptr())
        ctx.sync()
        ctx.set_stream(None)
    flags = np.zeros(parallel.words_for(cap) * 64, dtype=np.uint8)
    flags[:n_valid] = ho[:n_valid]
    flags[cap:cap + n_valid] = hi[:n_valid]
    assert np.array_equal(words.cpu().numpy(), np.packbits(flags, bitorder="little").view(np.int64))


def test_polygon_point_flag_grid_follows_radius_and_list_changes(oracle):
    """The flag-only point check looks at the obstacles of its grid cell only; the grid is padded for the robot radius
    (and the ball radius of a fused extend call) it was built for, and is rebuilt when a call needs more, when the pad
    is far too wide, and when the obstacle list changes.  A sequence of calls that forces all three, each against the
    call that also asks for the certificate (which walks the whole list in the reference's order)."""
    from rrtqx_3d_amd import _capi
    rng = np.random.default_rng(79)
    polys = synth.polygons(200)
    n = 4001
    P = np.zeros((n, 3))
    P[:, :2] = rng.uniform(-60, 60, (n, 2))
    with Context(3) as ctx:
        ctx.nodes_append(synth.nodes(20_000, 3))
        ctx.polygons_set(polys)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)

        def check(rr):
            full, _ = ctx.points_check(P, rr, kind=1)
            flag, _ = ctx.points_check(P, rr, kind=1, want_clearance=False)
            assert np.array_equal(full, flag), rr
            return int(full.sum())

        counts = [check(rr) for rr in (0.1, 3.0, 0.2, 25.0, 0.0, -0.5, 0.3)]
        assert counts[1] > counts[0] and counts[3] > counts[1]
        Q = synth.queries(512, 3)
        a = ctx.extend_candidates(Q, 6.0, 0.5)             # per-sample lists: the pad now covers the ball radius too
        check(0.5)
        ctx.polygons_set(polys[:90])                       # the list changes: records, grid and lists are rebuilt
        b = ctx.extend_candidates(Q, 6.0, 0.5)
        assert np.array_equal(a["idx"], b["idx"]) and not np.array_equal(a["hit_out"], b["hit_out"])
        ps = oracle.PolygonSet(polys[:90])
        for i in range(0, n, 97):
            u, _ = oracle.point_check_polygons(ps, P[i], 0.5)
            assert ctx.points_check(P[i:i + 300], 0.5, kind=1, want_clearance=False)[0][0] == u
        check(0.5)
