"""The reference-named host interface (rrtqx_3d_amd/drrt.py) on the GPU, written the way the
reference's own (commented) tests are: R/kdTree_general.jl:1039-1087 `testCase` inserts random
points and compares kdFindNearest / kdFindWithinRange with the naive versions; :1089-1148
`testGhost` does the same on R^2 x S^1.  The naive side here is the CPU oracle."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_testCase_kd_vs_naive(oracle):
    from rrtqx_3d_amd.drrt import (KDTree, RRTNode, emptyRangeList, kdFindNearest, kdFindWithinRange, kdInsert,
                                   kdInsertMany, popFromRangeList)
    rng = np.random.default_rng(0)
    d = 3
    pts = rng.random((5000, d))
    T = KDTree(d)
    ref = oracle.KDTree(d)
    kdInsert(T, RRTNode(pts[0]))                      # one-at-a-time insert, as the planner does
    kdInsertMany(T, [RRTNode(p) for p in pts[1:]])    # and the batched form
    ref.insert_many(pts)
    assert T.treeSize == 5000 and T.root.index == 0
    for query in rng.random((50, d)):
        node, dist = kdFindNearest(T, query)
        ni, nd = ref.nearest(query, naive=True)
        assert node.index == ni and dist == nd
        L = kdFindWithinRange(T, 0.1, query)
        nidx, nkey = ref.range_naive(0.1, query)
        assert L.length == len(nidx)
        got = {n.data.index: n.key for n in L}
        assert got == dict(zip(nidx.tolist(), nkey.tolist()))
        assert all(n.data.inHeap for n in L)
        if L.length:
            node, key = popFromRangeList(L)
            assert not node.inHeap
        emptyRangeList(L)                             # the caller must clear the flags
        assert L.length == 0 and not any(n.inHeap for n in T.nodes)


def test_find_more_within_range_merges(oracle):
    from rrtqx_3d_amd.drrt import KDTree, RRTNode, emptyRangeList, kdFindMoreWithinRange, kdFindWithinRange, kdInsert
    T = KDTree(3)
    for p in [[0, 0, 0], [1, 0, 0], [3, 0, 0]]:
        kdInsert(T, RRTNode(p))
    L = kdFindWithinRange(T, 1.5, [0, 0, 0])
    kdFindMoreWithinRange(T, 2.5, [2, 0, 0], L)
    got = {n.data.index: n.key for n in L}
    assert got == {0: 0.0, 1: 1.0, 2: 1.0}            # node 1 keeps the key of its first discovery
    emptyRangeList(L)


def test_testGhost_wrapped_theta(oracle):
    from rrtqx_3d_amd.drrt import KDTree, RRTNode, emptyRangeList, kdFindWithinRange, kdInsertMany
    rng = np.random.default_rng(1)
    two_pi = 2.0 * math.pi
    pts = np.concatenate([rng.uniform(-5, 5, (4000, 2)), np.zeros((4000, 1)), rng.uniform(0, two_pi, (4000, 1))], 1)
    T = KDTree(4, None, [4], [two_pi])                # wraps are 1-based dimension numbers (R/DRRT.jl:3312)
    ref = oracle.KDTree(4, wraps=[3], wrap_points=[two_pi])
    kdInsertMany(T, [RRTNode(p) for p in pts])
    ref.insert_many(pts)
    for q in pts[:40] + 0.01:
        L = kdFindWithinRange(T, 2.0, q)
        nidx, nkey = ref.range_naive(2.0, q)
        assert {n.data.index: n.key for n in L} == dict(zip(nidx.tolist(), nkey.tolist()))
        emptyRangeList(L)


def test_cspace_edge_and_point_checks(oracle):
    from rrtqx_3d_amd import synth
    from rrtqx_3d_amd.drrt import (CSpace, RRTNode, SphereObstacle, addObsToCSpace, calculateTrajectory,
                                   explicitEdgeCheck, explicitEdgeChecks, explicitNodeCheck, newEdge, validMove)
    S = CSpace(3, -1.0, [-50] * 3, [50] * 3, [0] * 3, [1] * 3)
    S.robotRadius = 0.5
    sph = synth.spheres(16)
    for row in sph:
        addObsToCSpace(S, SphereObstacle(row[:3], row[3]))        # listPush: list order = reverse of this loop
    list_order = sph[::-1]
    osph, m = oracle.make_spheres(list_order)
    rng = np.random.default_rng(2)
    a = [RRTNode(p) for p in rng.uniform(-50, 50, (300, 3))]
    b = [RRTNode(n.position + rng.normal(0, 4, (1, 3))) for n in a]
    edges = [newEdge(x, y) for x, y in zip(a, b)]
    hits = explicitEdgeChecks(S, edges)
    for e, h in zip(edges, hits):
        rh, _ = oracle.edge_check_spheres(osph, m, e.startNode.position, e.endNode.position, 0.5)
        assert bool(h) == rh
    e = edges[0]
    calculateTrajectory(S, e)
    assert e.dist == oracle.euclid(e.startNode.position, e.endNode.position) == e.distOriginal == e.Wdist
    assert validMove(S, e)
    # explicitEdgeCheck(S, edge, ob): one obstacle
    ob = list(S.obstacles)[3]
    one, m1 = oracle.make_spheres(list_order[3:4])
    for e in edges[:50]:
        assert explicitEdgeCheck(S, e, ob) == oracle.edge_check_spheres(one, m1, e.startNode.position,
                                                                        e.endNode.position, 0.5)[0]
    # obstacleUnused / warm-up behave like the reference (R/DRRT_Q.jl:1777, 1805)
    ob.obstacleUnused = True
    assert not any(explicitEdgeCheck(S, e, ob) for e in edges[:50])
    u, c = explicitNodeCheck(S, a[0])
    act = np.ones(16, dtype=np.uint8)
    act[3] = 0
    o2, m2 = oracle.make_spheres(list_order, act)
    assert (u, c) == oracle.point_check_spheres(o2, m2, a[0].position, 0.5, quick=True)
    S.inWarmupTime = True
    assert not explicitEdgeChecks(S, edges).any()
    assert explicitNodeCheck(S, a[0]) == (False, float("inf"))


def test_error_idiom():
    from rrtqx_3d_amd.drrt import KDTree, kdFindNearest
    T = KDTree(3)
    with pytest.raises(RuntimeError):      # the reference error()s; an empty tree is a state error here
        kdFindNearest(T, [0, 0, 0])


def test_obstacle_sweeps(oracle):
    """addNewObstacle / removeObstacle edge loops (R/DRRT_Q.jl:3220-3362) as batched calls."""
    from rrtqx_3d_amd import synth
    from rrtqx_3d_amd.drrt import (CSpace, KDTree, RRTNode, SphereObstacle, addObsToCSpace, emptyRangeList,
                                   findPointsInConflictWithObstacle, kdInsertMany, newEdge, obstacleSweepEdgeChecks)
    rng = np.random.default_rng(3)
    pts = rng.uniform(-20, 20, (6000, 3))               # the live script's 40^3 world
    KD = KDTree(3)
    nodes = [RRTNode(p) for p in pts]
    kdInsertMany(KD, nodes)
    S = CSpace(3, -1.0, [-20] * 3, [20] * 3, [0] * 3, [1] * 3)
    S.robotRadius, S.delta = 0.5, 8.0
    sph = np.concatenate([rng.uniform(-20, 20, (12, 3)), rng.uniform(1.0, 3.5, (12, 1))], 1)
    obs = [SphereObstacle(r[:3], r[3]) for r in sph]
    for o in obs:
        addObsToCSpace(S, o)
    obs[5].obstacleUnused = True
    obs[7].startTime, obs[7].lifeSpan = 100.0, 5.0       # outside its time window at t = 10
    new_ob = obs[2]
    # nodes in conflict: radius query around the obstacle centre
    L = findPointsInConflictWithObstacle(S, KD, new_ob)
    ref_tree = oracle.KDTree(3)
    ref_tree.insert_many(pts)
    ridx, _ = ref_tree.range_naive(0.5 + 8.0 + new_ob.radius, new_ob.position)
    conflict = sorted(n.data.index for n in L)
    assert conflict == ridx.tolist() and len(conflict) > 50
    emptyRangeList(L)
    # graph edges: each conflict node to 6 random other nodes
    edges = [newEdge(nodes[i], nodes[int(j)]) for i in conflict for j in rng.integers(0, 6000, 6)]
    # addNewObstacle: explicitEdgeCheck(S, edge, ob)
    hit = obstacleSweepEdgeChecks(S, KD, edges, ob=new_ob)
    one, m1 = oracle.make_spheres(sph[2:3])
    for e, h in zip(edges, hit):
        assert bool(h) == oracle.edge_check_spheres(one, m1, e.startNode.position, e.endNode.position, 0.5)[0]
    assert hit.any() and not hit.all()
    # removeObstacle: conflictsWithOtherObs over the obstacles active in their time window
    others = obstacleSweepEdgeChecks(S, KD, edges, others_of=new_ob, timeElapsed=10.0)
    list_order = list(S.obstacles)
    keep = [o for o in list_order if o is not new_ob and not o.obstacleUnused
            and o.startTime <= 10.0 <= o.startTime + o.lifeSpan]
    assert len(keep) == 9
    ko, mk = oracle.make_spheres(np.array([[*o.position, o.radius] for o in keep]))
    for e, h in zip(edges, others):
        assert bool(h) == oracle.edge_check_spheres(ko, mk, e.startNode.position, e.endNode.position, 0.5)[0]


@pytest.mark.gpu
def test_obstacle_sweep_polygon_through_the_mirror(oracle):
    """drrt.obstacleSweep with an Obstacle (polygon list): addNewObstacle's and removeObstacle's edge loops of the
    legacy planner (R/DRRT.jl:3127-3290) in one call each, against the oracle's restatement over the same edges."""
    import math
    from rrtqx_3d_amd import drrt
    rng = np.random.default_rng(17)
    S = drrt.CSpace(4, -1.0, [-15, -15, 0, 0], [15, 15, 0, 2 * math.pi], [0, 0, 0, 0], [1, 1, 0, 0])
    S.robotRadius, S.delta, S.minTurningRadius, S.spaceHasTheta = 0.5, 6.0, 1.0, True
    KD = drrt.KDTree(4, None, [4], [2.0 * math.pi])
    pts = np.c_[rng.uniform(-15, 15, (600, 2)), np.zeros(600), rng.uniform(0, 2 * math.pi, 600)]
    nodes = [drrt.RRTNode(p) for p in pts]
    for nd in nodes:
        drrt.kdInsert(KD, nd)
    squares = [np.array([[x, y], [x + 3, y], [x + 3, y + 2], [x, y + 2]], dtype=float) for x, y in ((-6, -4), (2, 1), (2.5, 1.5))]
    obs = [drrt.Obstacle(3, sq) for sq in squares]
    for ob in obs:
        ob.obstacleUnused = False
        drrt.listPush(S.obstacles, ob)
    edges = []
    for i in range(600):
        for j in rng.choice(600, 5, replace=False):
            if i != j:
                edges.append(drrt.newEdge(nodes[i], nodes[int(j)], drrt.DubinsEdge))
    assert drrt.registerEdges(KD, edges) == 0
    es = np.array([e.startNode.index for e in edges], dtype=np.int32)
    ee = np.array([e.endNode.index for e in edges], dtype=np.int32)
    tree = oracle.KDTree(4, wraps=[3], wrap_points=[2.0 * math.pi])
    tree.insert_many(pts)
    order = list(S.obstacles)                                   # list order: front = most recently pushed
    ps = oracle.PolygonSet([o.polygon for o in order])
    blocked = np.zeros(len(edges), dtype=bool)
    for ob in obs:
        j = order.index(ob)
        ids = drrt.obstacleSweep(S, KD, ob)
        want = oracle.add_new_obstacle_edges(tree, pts, es, ee, ps, j, 0.5, 6.0, dubins=True, r_min=1.0)
        assert np.array_equal(ids, want)
        blocked[ids] = True
    assert blocked.sum() > 20
    drrt.blockEdges(KD, np.nonzero(blocked)[0].astype(np.int32))
    dist = np.where(blocked, np.inf, 1.0)
    ob = obs[1]                                                 # overlaps obs[2]: shared edges stay blocked
    ids = drrt.obstacleSweep(S, KD, ob, remove=True)
    want = oracle.remove_obstacle_edges(tree, pts, es, ee, dist, ps, order.index(ob), 0.5, 6.0, dubins=True, r_min=1.0)
    assert np.array_equal(ids, want) and 0 < len(ids) < blocked.sum()
