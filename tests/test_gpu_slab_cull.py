"""Slab-culled range scan (RRTX_OPT_NN_CULL): skipping node chunks by their x extent must not
change a single neighbour index or stored distance.  Every case is checked against the CPU
oracle's kd-tree (R/kdTree_general.jl semantics) and against the unculled scan."""
import math

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu


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


def _both_modes(ctx, Q, r, ref):
    units = {}
    for mode in (2, 0):
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, mode)
        offsets, idx, dist = ctx.nn_radius(Q, r)
        _check_csr(offsets, idx, dist, ref)
        units[mode] = ctx.stats().last_scan_units
    ctx.set_option(_capi.RRTX_OPT_NN_CULL, 2)
    ctx.nn_radius(Q[:1], r if np.isscalar(r) else r[:1])     # so that stats() describe a culled call
    ctx.set_option(_capi.RRTX_OPT_NN_CULL, 1)
    assert units[0] == 0
    return units[2]


def test_cull_skips_most_chunks_and_matches(oracle):
    n, nq = 60_000, 2048
    pts = synth.nodes(n, 3)
    Q = synth.queries(nq, 3)
    r = synth.ball_radius(n, 3)
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        units = _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))
        tile = ctx.stats().last_tile_q
        n_tiles, n_chunks = (nq + tile - 1) // tile, (n + 511) // 512
        assert 0 < units < 0.35 * n_tiles * n_chunks     # x reach ~ (tile extent + 2 r) / 100 of the cloud


def test_cull_tail_rebuild_and_capacity_growth(oracle):
    """nodes appended after a rebuild live in an unsorted tail (chunk extents grow by atomics);
    enough of them trigger the next rebuild; the node store is reallocated on the way"""
    rng = np.random.default_rng(5)
    pts = rng.uniform(-50, 50, (30_000, 3))
    Q = rng.uniform(-55, 55, (700, 3))
    r = 4.0
    tree = oracle.KDTree(3)
    with Context(3, node_capacity=1024) as ctx:
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 2)
        done = 0
        for upto in (1, 5, 600, 9000, 9001, 9700, 12_000, 12_300, 30_000):
            ctx.nodes_append(pts[done:upto])
            tree.insert_many(pts[done:upto])
            done = upto
            offsets, idx, dist = ctx.nn_radius(Q, r)
            _check_csr(offsets, idx, dist, _oracle_lists(tree, Q, r))


def test_cull_wrapped_first_dimension_and_theta(oracle):
    """ghost copies of a wrapped x move to another bucket / slab; theta wrap leaves x alone"""
    rng = np.random.default_rng(17)
    n = 6000
    pts = rng.uniform(0.0, 1.0, (n, 3))
    pts[:, 1] *= 10.0
    tree = oracle.KDTree(3, wraps=[0, 2], wrap_points=[1.0, 1.0])
    tree.insert_many(pts)
    Q = pts[rng.integers(0, n, 128)] + rng.normal(0, 0.01, (128, 3))
    Q[:, [0, 2]] = np.clip(Q[:, [0, 2]], 0.0, 1.0)
    with Context(3) as ctx:
        ctx.set_wrap(0, 1.0)
        ctx.set_wrap(2, 1.0)
        ctx.nodes_append(pts)
        for r in (0.08, 0.35, 0.7):
            _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))
    n = 12_000
    pts = synth.nodes(n, 4)
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    Q = synth.queries(200, 4)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2.0 * math.pi)
        ctx.nodes_append(pts)
        for r in (10.0, 2.5):
            _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))


def test_cull_degenerate_extents_and_nonfinite(oracle):
    rng = np.random.default_rng(2)
    cases = []
    # every node on one x plane (zero-width extent: a single slab)
    p = rng.uniform(-5, 5, (3000, 3)); p[:, 0] = 1.25
    cases.append((p, rng.uniform(-5, 5, (100, 3)), 1.5))
    # two far apart clusters: most slabs empty
    p = np.concatenate([rng.normal(0, 1, (2000, 3)), rng.normal(0, 1, (2000, 3)) + [1e6, 0, 0]])
    q = np.concatenate([rng.normal(0, 1, (50, 3)), rng.normal(0, 1, (50, 3)) + [1e6, 0, 0], [[5e5, 0, 0]]])
    cases.append((p, q, 0.75))
    # inf / NaN / huge coordinates among the nodes and the queries
    p = rng.uniform(-5, 5, (1500, 3))
    p[7] = [np.inf, 0, 0]; p[8] = [np.nan, 0, 0]; p[9] = [-np.inf, 1, 1]; p[10] = [1e300, 0, 0]; p[11] = [0, np.nan, 0]
    q = rng.uniform(-5, 5, (40, 3))
    q[0] = [np.nan, 0, 0]; q[1] = [np.inf, 0, 0]; q[2] = [1e300, 0.5, 0]; q[3] = [0, np.inf, 0]
    cases.append((p, q, 2.0))
    for pts, Q, r in cases:
        tree = oracle.KDTree(3)
        tree.insert_many(pts)
        with Context(3) as ctx:
            ctx.nodes_append(pts)
            _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))


def test_cull_per_query_radii_on_the_slab_boundary(oracle):
    """radii from empty to everything, and nodes whose x distance equals the radius to the ulp
    (the chunk test is on x alone, so these sit exactly on the edge of a tile's reach)"""
    rng = np.random.default_rng(11)
    n = 20_000
    pts = rng.uniform(-50, 50, (n, 3))
    Q = rng.uniform(-50, 50, (300, 3))
    r = rng.uniform(0.0, 9.0, 300)
    r[0] = 0.0; r[1] = 1e-300; r[2] = 500.0; r[3] = np.inf
    for k in range(4, 120):          # node straight along x at distance r(1 + few ulps)
        rel = rng.choice([0.0, 1e-16, -1e-16, 2e-16, -2e-16, 1e-15, -1e-15])
        pts[k] = Q[k] + [r[k] * (1.0 + rel) * rng.choice([-1.0, 1.0]), 0.0, 0.0]
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))


def test_cull_third_coordinate_bins(oracle):
    """the sorted part is also cut by bins of the third coordinate: nodes straight above / below a query at the
    radius to the ulp (the edge of a tile's reach in z), flat and nearly flat trees, non-finite third coordinates,
    a tree that is a thin tall column"""
    rng = np.random.default_rng(12)
    cases = []
    n = 24_000
    pts = rng.uniform(-50, 50, (n, 3))
    Q = rng.uniform(-50, 50, (400, 3))
    r = rng.uniform(0.5, 9.0, 400)
    for k in range(4, 200):
        rel = rng.choice([0.0, 1e-16, -1e-16, 2e-16, -2e-16, 1e-15, -1e-15])
        pts[k] = Q[k] + [0.0, 0.0, r[k] * (1.0 + rel) * rng.choice([-1.0, 1.0])]
    cases.append((pts, Q, r))
    # flat (one bin), nearly flat (extent far below the (x, y) extent: the copies are ordered in (x, y) only)
    p = rng.uniform(-50, 50, (n, 3)); p[:, 2] = 3.0
    cases.append((p, np.concatenate([rng.uniform(-50, 50, (200, 2)), rng.uniform(0, 6, (200, 1))], axis=1), 4.0))
    p = rng.uniform(-50, 50, (n, 3)); p[:, 2] = rng.uniform(0, 1.0, n)
    cases.append((p, np.concatenate([rng.uniform(-50, 50, (200, 2)), rng.uniform(-2, 3, (200, 1))], axis=1), 2.5))
    # inf / NaN / huge third coordinates among nodes and queries
    p = rng.uniform(-20, 20, (n, 3))
    p[70, 2] = np.inf; p[80, 2] = np.nan; p[90, 2] = -np.inf; p[100, 2] = 1e300; p[110, 2] = -1e300
    q = rng.uniform(-20, 20, (300, 3))
    q[0, 2] = np.nan; q[1, 2] = np.inf; q[2, 2] = 1e300; q[3, 2] = -np.inf
    cases.append((p, q, 3.0))
    # a column: tiny (x, y) extent, long in z
    p = np.concatenate([rng.uniform(-0.5, 0.5, (n, 2)), rng.uniform(-500, 500, (n, 1))], axis=1)
    cases.append((p, np.concatenate([rng.uniform(-1, 1, (300, 2)), rng.uniform(-520, 520, (300, 1))], axis=1), 1.5))
    for pts, Q, r in cases:
        tree = oracle.KDTree(3)
        tree.insert_many(pts)
        with Context(3) as ctx:
            ctx.nodes_append(pts)
            units = _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))
            assert units > 0


def test_cull_more_than_1024_chunks():
    """a tree of more than 1024 chunks (the chunk list of a tile is built in windows) searched with the sorted
    part listed as groups, before and after an appended batch: culled == brute force"""
    n, nq = 560_000, 3000
    rng = np.random.default_rng(13)
    pts = rng.uniform(-70, 70, (n + 20_000, 3))
    Q = rng.uniform(-70, 70, (nq, 3))
    with Context(3) as ctx:
        ctx.nodes_append(pts[:n])
        for stage in range(2):
            outs = []
            for mode in (2, 0):
                ctx.set_option(_capi.RRTX_OPT_NN_CULL, mode)
                outs.append(ctx.nn_radius(Q, 3.0))
            for a, b in zip(*outs):
                assert np.array_equal(a, b)
            assert outs[0][0][-1] > nq
            ctx.nodes_append(pts[n:])          # a sorted run behind more than 1024 chunks


def test_cull_extend_candidates_same_as_unculled():
    n, nq = 40_000, 1500
    pts, Q, sph = synth.nodes(n, 3), synth.queries(nq, 3), synth.spheres(64)
    r = synth.ball_radius(n, 3)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph)
        outs = []
        for mode in (2, 0):
            ctx.set_option(_capi.RRTX_OPT_NN_CULL, mode)
            outs.append(ctx.extend_candidates(Q, r, 0.5))
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), k
        assert len(outs[0]["idx"]) > nq


def test_cull_extend_samples_checked_when_no_ball_can_hold_a_node():
    """explicitPointCheck of a sample does not depend on its ball: whole tiles of samples that cannot have a
    neighbour (NaN coordinates sort into one bucket; r = 0) still get their collision flag in the fused path."""
    n, nq = 40_000, 600
    pts, sph = synth.nodes(n, 3), synth.spheres(64)
    Q = synth.queries(nq, 3)
    Q[:200] = sph[np.arange(200) % 64, :3] + 0.01          # inside a sphere
    Q[200:260, 0] = np.nan                                   # NaN sample: !(s >= thr) holds, so "in collision"
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph)
        for r in (0.0, synth.ball_radius(n, 3)):
            outs = []
            for mode in (2, 0):
                ctx.set_option(_capi.RRTX_OPT_NN_CULL, mode)
                outs.append(ctx.extend_candidates(Q, r, 0.5))
            for k in outs[0]:
                assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), (r, k)
            assert outs[0]["sample_unsafe"][:260].all() and not outs[0]["sample_unsafe"].all()


def test_cull_every_lane_flagged_and_list_overflow(oracle):
    """balls that swallow the whole tree: every lane of every copy files an entry (the waves must
    confirm their own slices on the way), the per-copy LDS lists and the per-query buckets overflow
    into the shared list, lists are far longer than one wave"""
    rng = np.random.default_rng(21)
    n, nq = 9000, 48
    pts = rng.uniform(-10, 10, (n, 3))
    Q = rng.uniform(-10, 10, (nq, 3))
    r = np.full(nq, 1e3)
    r[::5] = 6.0                     # a few ordinary balls in between
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ref = _oracle_lists(tree, Q, r)
        assert sum(len(i) for i, _ in ref) > 300_000
        _both_modes(ctx, Q, r, ref)


def test_cull_large_batch_walks_tiles_and_block_sums():
    """more tiles than resident workgroups (a workgroup walks several tiles and reuses its LDS and
    slices) and more than 65536 queries (two-level offsets scan); checked against the unculled scan"""
    n, nq = 30_000, 70_001
    pts = synth.nodes(n, 3)
    rng = np.random.default_rng(4)
    Q = rng.uniform(-50, 50, (nq, 3))
    Q[1000:1200] = Q[0]              # many copies of one query: one crowded cell
    r = 2.5
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        outs = []
        for mode in (2, 0):
            ctx.set_option(_capi.RRTX_OPT_NN_CULL, mode)
            outs.append(ctx.nn_radius(Q, r))
        for a, b in zip(*outs):
            assert np.array_equal(a, b)
        assert outs[0][0][-1] > nq
        # spot check against numpy on a few queries (first-principles, same arithmetic)
        off, idx, dist = outs[0]
        for i in (0, 1100, 35_000, 70_000):
            d = pts - Q[i]
            s = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
            want = np.nonzero(np.sqrt(s) < r)[0]
            if np.sqrt(s[0]) <= r and 0 not in want:
                want = np.concatenate([[0], want])
            assert np.array_equal(idx[off[i]:off[i + 1]], want)
            assert np.array_equal(dist[off[i]:off[i + 1]], np.sqrt(s[want]))


@pytest.mark.parametrize("seed", range(6))
def test_cull_randomised_scenarios(oracle, seed):
    """random trees (uniform / clustered / duplicated points / lattice), appended in random pieces
    with searches in between, random per-query radii, with and without a wrapped dimension:
    culled == brute force == oracle after every piece"""
    rng = np.random.default_rng(1000 + seed)
    d = 3 if seed % 2 == 0 else 4
    n = int(rng.integers(2_000, 14_000))
    kind = seed % 4
    if kind == 0:
        pts = rng.uniform(-30, 30, (n, d))
    elif kind == 1:                                   # a few tight clusters + background
        c = rng.uniform(-30, 30, (5, d))
        pts = c[rng.integers(0, 5, n)] + rng.normal(0, 0.7, (n, d))
        pts[: n // 10] = rng.uniform(-30, 30, (n // 10, d))
    elif kind == 2:                                   # many exact duplicates
        base = rng.uniform(-30, 30, (n // 8 + 1, d))
        pts = base[rng.integers(0, len(base), n)]
    else:                                             # integer lattice: ties in x and y everywhere
        pts = rng.integers(-12, 13, (n, d)).astype(np.float64)
    wraps, wrap_points = (None, None)
    if d == 4:
        pts[:, 3] = rng.uniform(0, 2 * math.pi, n)
        pts[:, 2] = 0.0
        wraps, wrap_points = [3], [2 * math.pi]
    nq = int(rng.integers(1, 400))
    Q = pts[rng.integers(0, n, nq)] + rng.normal(0, 1.5, (nq, d))
    if d == 4:
        Q[:, 3] = np.mod(Q[:, 3], 2 * math.pi)
    r = rng.uniform(0.0, 6.0, nq)
    tree = oracle.KDTree(d, wraps=wraps, wrap_points=wrap_points)
    with Context(d, node_capacity=1024) as ctx:
        if d == 4:
            ctx.set_wrap(3, 2 * math.pi)
        cuts = sorted(set([n] + [int(x) for x in rng.integers(1, n, 3)]))
        done = 0
        for upto in cuts:
            ctx.nodes_append(pts[done:upto])
            tree.insert_many(pts[done:upto])
            done = upto
            _both_modes(ctx, Q, r, _oracle_lists(tree, Q, r))


def test_cull_sorted_runs_between_rebuilds(oracle):
    """batches appended after the index exists are laid out in cell order inside their own range of positions
    (sorted runs: strips of the grid instead of samples of the world).  Range search, nearest (empty balls) and
    the fused extend() preamble must not notice: same lists as the oracle, same as the unculled scan, and the
    runs' chunks are indeed skipped by most tiles."""
    rng = np.random.default_rng(77)
    n0, nb, steps = 40_000, 6000, 5
    pts = rng.uniform(-50, 50, (n0 + nb * steps, 3))
    pts[n0 + 100] = [np.nan, 0.0, 0.0]                    # a node nothing can find, inside a run
    pts[n0 + 101] = pts[17]                               # a duplicate of an indexed node
    Q = np.concatenate([rng.uniform(-50, 50, (1500, 3)), rng.uniform(60, 70, (4, 3))])     # the last four: empty balls
    r = 4.5
    sph = np.concatenate([rng.uniform(-40, 40, (24, 3)), rng.uniform(1.0, 3.5, (24, 1))], 1)
    tree = oracle.KDTree(3)
    tree.insert_many(pts[:n0])
    pts_o = pts.copy()
    pts_o[n0 + 100] = 1e9                                 # the oracle's stand-in: as unreachable, keeps indices aligned
    with Context(3) as ctx:
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 2)
        ctx.spheres_set(sph, np.ones(24, dtype=np.uint8))
        ctx.nodes_append(pts[:n0])
        ctx.nn_radius(Q[:8], r)                           # builds the index
        units_runs = []
        for s in range(steps):
            a, b = n0 + s * nb, n0 + (s + 1) * nb
            ctx.nodes_append(pts[a:b])
            tree.insert_many(pts_o[a:b])
            ref = _oracle_lists(tree, Q, r)
            offsets, idx, dist = ctx.nn_radius(Q, r)
            _check_csr(offsets, idx, dist, ref)
            units_runs.append(ctx.stats().last_scan_units)
            ctx.set_option(_capi.RRTX_OPT_NN_CULL, 0)
            o2, i2, d2 = ctx.nn_radius(Q, r)
            ctx.set_option(_capi.RRTX_OPT_NN_CULL, 2)
            assert np.array_equal(offsets, o2) and np.array_equal(idx, i2) and np.array_equal(dist, d2)
            near_idx, near_d = ctx.nn_nearest(Q[-4:])
            for k in range(4):
                want_i, want_d = tree.nearest(Q[-4 + k])
                assert near_idx[k] == want_i and near_d[k] == want_d
        # the fused extend() preamble over the runs == the unfused one over the unculled scan
        ext = ctx.extend_candidates(Q, r, 0.5)
        ctx.set_option(_capi.RRTX_OPT_NN_CULL, 0)
        ext0 = ctx.extend_candidates(Q, r, 0.5)
        assert ext.keys() == ext0.keys() and len(ext["idx"]) > 1000
        for key in ext:
            assert np.array_equal(ext[key], ext0[key]), key
        # a run's chunks are strips: most tiles skip them (unsorted, every tile screens every tail chunk)
        n_tiles = (len(Q) + 15) // 16
        tail_chunks = (nb * steps) // 512
        assert units_runs[-1] < 0.5 * n_tiles * tail_chunks
