"""The cost-propagation oracle (oracle/rrtx_oracle_graph.c: rewire / reduceInconsistency / propogateDescendants,
R/DRRT_Q.jl:2490-2541, 2647-2817) pinned by hand-worked cases and by an independent shortest-path solver.

The reference holds no fixtures for these functions (it has no tests); the hand cases below are worked from its
source text, and the fixed-point property (changeThresh = 0, queue run dry => rrtLMC = cost of the cheapest
route to the root) is checked against scipy's Dijkstra on graphs with integer costs, where every sum is exact.
Beyond those: parity unpinned."""
import numpy as np
import pytest
import scipy.sparse as sp
from scipy.sparse.csgraph import dijkstra

from oracle import oracle as orc

INF = float("inf")


def _fresh(n_nodes, edges, root):
    """graph with every node at Inf except the root, which sits in the queue (the state after the root is added)"""
    g = orc.Graph(n_nodes + 1)                 # the extra node is an unreachable goal: the loop runs dry
    for (a, b, w) in edges:
        g.add_edge(a, b, w)
    for v in range(n_nodes + 1):
        g.set_node(v, INF, INF)
    g.set_node(root, 0.0, INF)
    g.verifyInQueue(root)
    return g


def test_chain_by_hand():
    # 2 -> 1 -> 0 (root), costs 1.5 and 1: lmc = [0, 1, 2.5]; parent edges are the only out-edges
    g = _fresh(3, [(1, 0, 1.0), (2, 1, 1.5)], 0)
    g.reduceInconsistency(3, 0)
    assert g.lmc()[:3].tolist() == [0.0, 1.0, 2.5]
    assert g.parent_edge()[:3].tolist() == [-1, 0, 1]
    assert g.queue_length() == 0
    assert g.tree_cost()[:3].tolist() == [0.0, 1.0, 2.5]


def test_detour_after_block_by_hand():
    # 2 has two ways home: through 1 (1.5 + 1) or straight (4).  Blocking 2 -> 1 orphans 2, which re-attaches
    # along the direct edge.
    g = _fresh(3, [(1, 0, 1.0), (2, 1, 1.5), (2, 0, 4.0)], 0)
    g.reduceInconsistency(3, 0)
    assert g.lmc()[:3].tolist() == [0.0, 1.0, 2.5]
    g.blockEdge(1)
    g.propogateDescendants()
    g.reduceInconsistency(3, 0)
    assert g.lmc()[:3].tolist() == [0.0, 1.0, 4.0]
    assert g.parent_edge()[2] == 2


def test_orphan_without_alternative_stays_inf():
    g = _fresh(4, [(1, 0, 1.0), (2, 1, 1.0), (3, 2, 1.0)], 0)
    g.reduceInconsistency(4, 0)
    assert g.lmc()[:4].tolist() == [0.0, 1.0, 2.0, 3.0]
    g.blockEdge(1)                              # 2 -> 1: nodes 2 and 3 lose the root
    g.propogateDescendants()
    g.reduceInconsistency(4, 0)
    lmc = g.lmc()
    assert lmc[0] == 0.0 and lmc[1] == 1.0 and lmc[2] == INF and lmc[3] == INF
    assert g.parent_edge()[2] == -1 and g.parent_edge()[3] == -1


def _random_graph(rng, n, deg, integer=True):
    """both directions of every link, as extend() creates them; one link per pair of nodes (the planner never
    holds two edges between the same ordered pair, and rewire's "already my parent" test relies on that)"""
    a = np.repeat(np.arange(n), deg)
    b = rng.integers(0, n, size=n * deg)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    _, first = np.unique(lo * n + hi, return_index=True)
    first = first[lo[first] != hi[first]]
    a, b = lo[first], hi[first]
    w = rng.integers(1, 50, size=a.shape[0]).astype(np.float64) if integer else rng.uniform(0.1, 5.0, size=a.shape[0])
    s = np.concatenate([a, b])
    e = np.concatenate([b, a])
    ww = np.concatenate([w, w])
    return s, e, ww


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fixed_point_is_shortest_route(seed):
    rng = np.random.default_rng(seed)
    n = 400
    s, e, w = _random_graph(rng, n, 3)
    g = _fresh(n, list(zip(s.tolist(), e.tolist(), w.tolist())), 0)
    g.reduceInconsistency(n, 0)
    # route cost from v to the root along edges v -> u == dijkstra on the reversed graph from the root
    d = dijkstra(sp.coo_matrix((w, (e, s)), shape=(n, n)).tocsr(), indices=0)
    assert np.array_equal(g.lmc()[:n], d)
    # every parent edge attains its node's value
    par = g.parent_edge()[:n]
    lmc = g.lmc()
    for v in range(1, n):
        if np.isfinite(lmc[v]):
            assert s[par[v]] == v and lmc[e[par[v]]] + w[par[v]] == lmc[v]


def test_block_and_repair_matches_fresh_solve():
    rng = np.random.default_rng(7)
    n = 300
    s, e, w = _random_graph(rng, n, 3)
    edges = list(zip(s.tolist(), e.tolist(), w.tolist()))
    g = _fresh(n, edges, 0)
    g.reduceInconsistency(n, 0)
    blocked = rng.choice(len(edges), size=60, replace=False)
    for b in blocked:
        g.blockEdge(int(b))
    g.propogateDescendants()
    g.reduceInconsistency(n, 0)
    w2 = w.copy()
    w2[blocked] = INF
    h = _fresh(n, list(zip(s.tolist(), e.tolist(), w2.tolist())), 0)
    h.reduceInconsistency(n, 0)
    assert np.array_equal(g.lmc()[:n], h.lmc()[:n])
