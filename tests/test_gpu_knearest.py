"""rrtx_nn_knearest (kdFindKNearest, R/kdTree_general.jl:696-723) against the oracle's restatement of
the reference's kd-tree + max-heap search: same node set, bit-identical distances, through the C-ABI.
The GPU rows are sorted by (distance, index); the reference returns heap order, so sets are compared."""
import numpy as np
import pytest

from rrtqx_3d_amd import _capi, drrt
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu


def _check_rows(oracle_tree, pts, queries, k, idx, dist, count):
    n = len(pts)
    width = min(max(k, 2), n)
    assert idx.shape == (len(queries), max(k, 2))
    for i, q in enumerate(queries):
        c = int(count[i])
        assert c == width
        oi, ok = oracle_tree.knearest(k, q)
        o = np.argsort(oi)
        g = np.argsort(idx[i, :c])
        assert np.array_equal(idx[i, :c][g], oi[o])
        assert np.array_equal(dist[i, :c][g], ok[o])                   # bit-identical keys
        d = dist[i, :c]
        assert np.all(d[1:] >= d[:-1])                                 # ascending distance
        assert np.all(idx[i, c:] == -1) and np.all(np.isinf(dist[i, c:]))


@pytest.mark.parametrize("d", [3, 4])
def test_knearest_matches_reference_search(oracle, d):
    rng = np.random.default_rng(500 + d)
    pts = rng.uniform(-50, 50, (6000, d))
    queries = rng.uniform(-55, 55, (48, d))
    t = oracle.KDTree(d)
    t.insert_many(pts)
    with Context(d) as ctx:
        ctx.nodes_append(pts)
        for k in (1, 2, 5, 32, 257, 2048):
            idx, dist, count = ctx.nn_knearest(queries, k)
            _check_rows(t, pts, queries, k, idx, dist, count)


@pytest.mark.parametrize("n_cluster_queries", [3, 400])
def test_knearest_list_path_equals_exhaustive(oracle, n_cluster_queries):
    """Large batches on a big tree take the k nearest from range-search lists (radius guessed from a
    sample) and fall back to the exhaustive kernel per query; both must agree bit for bit.  The scene
    forces every branch: uniform background, a dense cluster (lists too long for the LDS sort, or --
    with many such queries -- more hits than the list buffer holds), queries far outside the cloud
    (lists too short) and a NaN query."""
    rng = np.random.default_rng(2024)
    pts = np.r_[rng.uniform(-50, 50, (40000, 3)), rng.normal(0, 0.5, (20000, 3)) + [10.0, 10.0, 10.0]]
    Q = np.r_[rng.uniform(-50, 50, (1500, 3)), rng.normal(0, 0.5, (n_cluster_queries, 3)) + [10.0, 10.0, 10.0],
              rng.uniform(200, 300, (60, 3)), [[np.nan, 0.0, 0.0]]]
    Q = Q[rng.permutation(len(Q))]
    t = oracle.KDTree(3)
    t.insert_many(pts)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        for k in (1, 16, 128, 129, 400, 513):
            ctx.set_option(_capi.RRTX_OPT_KNN_LISTS, 1)
            a = ctx.nn_knearest(Q, k)
            ctx.set_option(_capi.RRTX_OPT_KNN_LISTS, 0)
            b = ctx.nn_knearest(Q, k)
            for x, y in zip(a, b):
                assert np.array_equal(x, y, equal_nan=True)
            ok = ~np.isnan(Q[:, 0])
            assert (a[2][ok] == max(k, 2)).all() and (a[2][~ok] == 0).all()
            for i in rng.choice(np.flatnonzero(ok), 24, replace=False):
                oi, okey = t.knearest(k, Q[i])
                o, g = np.argsort(oi), np.argsort(a[0][i])
                assert np.array_equal(a[0][i][g], oi[o]) and np.array_equal(a[1][i][g], okey[o])


def test_knearest_small_trees_and_appends(oracle):
    rng = np.random.default_rng(77)
    pts = rng.uniform(-5, 5, (9, 3))
    t = oracle.KDTree(3)
    with Context(3) as ctx:
        for j, p in enumerate(pts):                                    # grows one node at a time
            t.insert(p)
            ctx.nodes_append([p])
            q = rng.uniform(-5, 5, (3, 3))
            for k in (1, 3, 20):
                idx, dist, count = ctx.nn_knearest(q, k)
                _check_rows(t, pts[:j + 1], q, k, idx, dist, count)


def test_knearest_ties_take_lowest_indices():
    # 600 nodes on 6 coincident sites: every distance is attained 100 times
    sites = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3], [4, 0, 0], [0, 5, 0]], dtype=np.float64)
    pts = np.tile(sites, (100, 1))
    q = np.array([[0.1, 0.0, 0.0]])
    s = ((pts - q[0]) ** 2)
    s = (s[:, 0] + s[:, 1]) + s[:, 2]
    order = np.lexsort((np.arange(len(pts)), s))                       # (distance, index)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        for k in (2, 7, 100, 101, 250, 600, 900):
            idx, dist, count = ctx.nn_knearest(q, k)
            c = min(k, 600)
            assert count[0] == c
            assert np.array_equal(idx[0, :c], order[:c])
            assert np.array_equal(dist[0, :c], np.sqrt(s[order[:c]]))


def test_knearest_skips_nonfinite_and_reports_errors():
    pts = np.array([[0, 0, 0], [1, 0, 0], [np.nan, 0, 0], [2, 0, 0], [np.inf, 0, 0]], dtype=np.float64)
    with Context(3) as ctx:
        with pytest.raises(_capi.RrtxError) as e:
            ctx.nn_knearest([[0, 0, 0]], 2)
        assert e.value.code == _capi.RRTX_E_STATE
        ctx.nodes_append(pts)
        idx, dist, count = ctx.nn_knearest([[0.4, 0, 0]], 5)
        assert count[0] == 3 and list(idx[0, :3]) == [0, 1, 3] and list(idx[0, 3:]) == [-1, -1]
        for bad in (0, -3, 2049):
            with pytest.raises(_capi.RrtxError) as e:
                ctx.nn_knearest([[0, 0, 0]], bad)
            assert e.value.code == _capi.RRTX_E_INVALID
        idx, dist, count = ctx.nn_knearest(np.zeros((0, 3)), 4)
        assert idx.shape == (0, 4)
    with Context(4) as ctx:                                            # wrapped space: the reference raises
        ctx.set_wrap(3, 2 * np.pi)
        ctx.nodes_append([[0, 0, 0, 1.0]])
        with pytest.raises(_capi.RrtxError) as e:
            ctx.nn_knearest([[0, 0, 0, 0.5]], 2)
        assert e.value.code == _capi.RRTX_E_STATE and "wrapped" in str(e.value)


def test_kdFindKNearest_mirror(oracle):
    rng = np.random.default_rng(9)
    pts = rng.uniform(-10, 10, (300, 3))
    tree = drrt.KDTree(3)
    t = oracle.KDTree(3)
    for p in pts:
        drrt.kdInsert(tree, drrt.RRTNode(p))
        t.insert(p)
    q = rng.uniform(-10, 10, 3)
    nodes = drrt.kdFindKNearest(tree, 6, q)
    oi, ok = t.knearest(6, q)
    assert sorted(n.index for n in nodes) == sorted(oi)
    assert sorted(n.data for n in nodes) == sorted(ok)
    assert len(drrt.kdFindKNearest(tree, 1, q)) == 2                   # the seed quirk
    n, d = drrt.kdFindNearestWithGuesstree(tree, q, tree.nodes[17])
    assert (n.index, d) == t.nearest(q)
