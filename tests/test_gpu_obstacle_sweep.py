"""addNewObstacle's edge loop against the device mirror of the planner's edges
(rrtx_graph_edges_append / rrtx_obstacle_sweep, R/DRRT_Q.jl:3195-3290) vs the oracle:
kdFindWithinRange around the obstacle (root with <=), every registered out-edge of the nodes
found, explicitEdgeCheck(S, edge, ob) against that one obstacle."""
import numpy as np
import pytest

from rrtqx_3d_amd import _capi, drrt
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
RR, DELTA = 0.5, 8.0


def _expected(oracle, pts, es, ee, sph_row, active, search_range):
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    idx, _ = tree.within_range(search_range, sph_row[:3])
    near = np.zeros(len(pts), dtype=bool)
    near[idx] = True
    cand = np.nonzero(near[es])[0]
    if len(cand) == 0 or not active:
        return np.zeros(0, dtype=np.int64)
    osph, m = oracle.make_spheres(sph_row[None, :])
    hit, _ = oracle.edges_check_spheres(osph, m, pts[es[cand]], pts[ee[cand]], RR)
    return cand[hit.astype(bool)]


@pytest.mark.parametrize("n", [3000, 40_000])
def test_obstacle_sweep_matches_oracle(oracle, n):
    rng = np.random.default_rng(n)
    pts = rng.uniform(-30, 30, (n, 3))
    # a k-nearest-ish random graph: every node gets 6 out-edges to nearby indices plus a "parent" edge
    es = np.repeat(np.arange(n), 7)
    ee = (es + rng.integers(1, 50, len(es))) % n
    ee[::7] = rng.integers(0, n, n)                    # long edges too
    es[:5], ee[:5] = 0, [1, 2, 3, 4, 5]                # out-edges of the root
    ee[5] = es[5]                                      # a zero-length edge: collides with any active obstacle in range
    sph = np.concatenate([rng.uniform(-25, 25, (12, 3)), rng.uniform(1.0, 6.0, (12, 1))], 1)
    sph[3, :3] = pts[0] + [2.0, 0.0, 0.0]              # an obstacle right at the root
    active = np.ones(12, dtype=np.uint8)
    active[7] = 0
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph, active)
        # registered in two pieces, ids are consecutive
        assert ctx.graph_edges_append(es[:1000], ee[:1000]) == 0
        assert ctx.graph_edges_append(es[1000:], ee[1000:]) == 1000
        assert ctx.n_graph_edges == len(es)
        total = 0
        for j in range(12):
            rng_j = RR + DELTA + sph[j, 3]
            got = ctx.obstacle_sweep(j, rng_j, RR, cap=16)          # small capacity: exercises the two-call path
            want = _expected(oracle, pts, es, ee, sph[j], active[j], rng_j)
            assert np.array_equal(got, want), j
            total += len(got)
        assert total > 50 and len(ctx.obstacle_sweep(7, 20.0, RR)) == 0
        # the root rule: a range that reaches the root exactly
        d0 = float(np.sqrt(((sph[3, :3] - pts[0]) ** 2).sum()))
        for r in (d0, np.nextafter(d0, 0)):
            got = ctx.obstacle_sweep(3, r, RR)
            assert np.array_equal(got, _expected(oracle, pts, es, ee, sph[3], 1, r))
        with pytest.raises(_capi.RrtxError):
            ctx.obstacle_sweep(99, 1.0, RR)
        with pytest.raises(_capi.RrtxError):
            ctx.graph_edges_append([0], [n])
        ctx.graph_edges_clear()
        assert ctx.n_graph_edges == 0 and len(ctx.obstacle_sweep(3, 30.0, RR)) == 0


def test_obstacle_sweep_through_the_mirror_names(oracle):
    rng = np.random.default_rng(5)
    KD = drrt.KDTree(3)
    S = drrt.CSpace(3, 0.0, [-20] * 3, [20] * 3, [0, 0, 0], [0, 0, 0])
    S.robotRadius, S.delta = RR, DELTA
    S.bind(KD)
    nodes = [drrt.RRTNode(p) for p in rng.uniform(-20, 20, (2000, 3))]
    drrt.kdInsertMany(KD, nodes)
    edges = [drrt.newEdge(nodes[i], nodes[int(j)]) for i in range(2000) for j in rng.integers(0, 2000, 3)]
    assert drrt.registerEdges(KD, edges) == 0
    ob = drrt.SphereObstacle([1.0, -2.0, 3.0, 4.0])
    drrt.addObsToCSpace(S, drrt.SphereObstacle([15.0, 15.0, 15.0, 1.0]))
    drrt.addObsToCSpace(S, ob)
    ids = drrt.obstacleSweep(S, KD, ob)
    # the same through the one-at-a-time names
    L = drrt.findPointsInConflictWithObstacle(S, KD, ob)
    near = {id(nd) for nd, _ in L.items()}
    want = [k for k, e in enumerate(edges) if id(e.startNode) in near and drrt.explicitEdgeCheck(S, e, ob)]
    assert list(ids) == want and len(want) > 10
