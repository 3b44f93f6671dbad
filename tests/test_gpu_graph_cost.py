"""Device cost propagation over the edge mirror (rrtx_graph_cost_to_root) against the oracle's
restatement of rewire / reduceInconsistency / propogateDescendants (R/DRRT_Q.jl:2490-2541, 2647-2817) run with
changeThresh = 0 until its queue is empty.  rrtLMC values are compared bit for bit; parent edges are compared
exactly where one edge alone attains the minimum (the reference breaks ties by visiting order)."""
import numpy as np
import pytest

from rrtqx_3d_amd import _capi
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
INF = float("inf")
RR = 0.5


def _geometric_graph(oracle, pts, r):
    """extend()'s graph: both directed edges between every pair of nodes closer than r (exact, from the oracle)"""
    tree = oracle.KDTree(pts.shape[1])
    tree.insert_many(pts)
    s, e = [], []
    for i in range(len(pts)):
        idx, _ = tree.within_range(r, pts[i])
        idx = np.asarray(idx)
        idx = idx[idx != i]
        s.append(np.full(len(idx), i))
        e.append(idx)
    return np.concatenate(s).astype(np.int32), np.concatenate(e).astype(np.int32)


def _edge_dist(pts, s, e):
    d = pts[s] - pts[e]
    acc = d[:, 0] * d[:, 0]
    for k in range(1, pts.shape[1]):
        acc = acc + d[:, k] * d[:, k]
    return np.sqrt(acc)


def _oracle_solve(oracle, n, s, e, w, root, blocked=()):
    """the reference's path to the same state: solve, then block + propogateDescendants + reduceInconsistency"""
    g = oracle.Graph(n + 1)                              # node n: a goal that stays at Inf, so the queue runs dry
    for a, b, c in zip(s.tolist(), e.tolist(), w.tolist()):
        g.add_edge(a, b, c)
    for v in range(n + 1):
        g.set_node(v, INF, INF)
    g.set_node(root, 0.0, INF)
    g.verifyInQueue(root)
    g.reduceInconsistency(n, root)
    if len(blocked):
        for b in blocked:
            g.blockEdge(int(b))
        g.propogateDescendants()
        g.reduceInconsistency(n, root)
    return g.lmc()[:n], g.parent_edge()[:n]


def _check_parents(lmc, par, s, e, w, root):
    n = len(lmc)
    assert par[root] == -1
    fin = np.isfinite(lmc)
    assert np.all(par[~fin] == -1)
    v = np.nonzero(fin)[0]
    v = v[v != root]
    assert np.all(par[v] >= 0)
    assert np.array_equal(s[par[v]], v)
    assert np.array_equal(lmc[e[par[v]]] + w[par[v]], lmc[v])
    # the lowest id among the edges that attain the value
    ok = np.isfinite(w) & fin[e] & fin[s]
    att = np.nonzero(ok & (np.where(ok, lmc[e] + np.where(ok, w, 0), INF) == lmc[s]) & (s != root))[0]
    lowest = np.full(n, np.iinfo(np.int64).max)
    np.minimum.at(lowest, s[att], att)
    assert np.array_equal(lowest[v], par[v])
    return att


@pytest.mark.parametrize("n,r", [(2000, 4.5), (20_000, 2.6)])
def test_cost_to_root_matches_reduce_inconsistency(oracle, n, r):
    rng = np.random.default_rng(n)
    pts = rng.uniform(-20, 20, (n, 3))
    s, e = _geometric_graph(oracle, pts, r)
    w = _edge_dist(pts, s, e)
    root = 17
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.graph_edges_append(s, e)                     # edge.dist defaults to the SimpleEdge cost
        lmc, par, passes = ctx.graph_cost_to_root(root)
        want, want_par = _oracle_solve(oracle, n, s, e, w, root)
        assert np.array_equal(lmc, want)
        assert lmc[root] == 0.0 and np.isfinite(lmc).sum() > n // 2 and passes >= 16
        att = _check_parents(lmc, par, s, e, w, root)
        single = np.bincount(s[att], minlength=n) == 1
        assert np.array_equal(par[single], want_par[single])

        # a new obstacle: the edges the sweep returns are blocked, orphans re-attach or stay at Inf
        sph = np.array([[pts[root, 0] + 3.0, pts[root, 1], pts[root, 2], 4.0], [5.0, 5.0, 5.0, 6.0]])
        ctx.spheres_set(sph, np.ones(2, dtype=np.uint8))
        blocked = np.concatenate([ctx.obstacle_sweep(j, RR + r + sph[j, 3], RR) for j in range(2)])
        blocked = np.unique(blocked)
        assert len(blocked) > 20
        ctx.graph_edges_block(blocked)
        lmc2, par2, _ = ctx.graph_cost_to_root(root)
        want2, want_par2 = _oracle_solve(oracle, n, s, e, w, root, blocked)
        assert np.array_equal(lmc2, want2)
        assert np.any(lmc2 != lmc) and np.all(lmc2 >= lmc)
        w2 = w.copy()
        w2[blocked] = INF
        att2 = _check_parents(lmc2, par2, s, e, w2, root)
        single2 = np.bincount(s[att2], minlength=n) == 1
        assert np.array_equal(par2[single2], want_par2[single2])
        # without the parent array
        lmc3, none, _ = ctx.graph_cost_to_root(root, want_parent=False)
        assert none is None and np.array_equal(lmc3, lmc2)


def test_costs_set_by_the_host_and_orphans(oracle):
    # costs that are not distances (Dubins lengths, costs with time), a part of the graph cut off from the root
    rng = np.random.default_rng(3)
    n = 5000
    pts = rng.uniform(-10, 10, (n, 3))
    a = np.repeat(np.arange(n), 4)
    b = (a + rng.integers(1, 40, len(a))) % n
    island = (a >= 4000) != (b >= 4000)                  # no edge crosses between [0, 4000) and [4000, n)
    a, b = a[~island], b[~island]
    key = np.minimum(a, b).astype(np.int64) * n + np.maximum(a, b)
    _, first = np.unique(key, return_index=True)
    a, b = a[first], b[first]
    s = np.concatenate([a, b]).astype(np.int32)
    e = np.concatenate([b, a]).astype(np.int32)
    w = rng.uniform(0.05, 9.0, len(s))
    w[rng.choice(len(s), 300, replace=False)] = INF     # edges validMove / explicitEdgeCheck already refused
    w[7] = 0.0                                           # a free edge
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        first_id = ctx.graph_edges_append(s[:3000], e[:3000])
        ctx.graph_edges_append(s[3000:], e[3000:])
        ctx.graph_edges_set_dist(first_id, w[:100])
        ctx.graph_edges_set_dist(100, w[100:])
        lmc, par, _ = ctx.graph_cost_to_root(0)
        want, _ = _oracle_solve(oracle, n, s, e, w, 0)
        assert np.array_equal(lmc, want)
        assert np.all(np.isinf(lmc[4000:])) and np.all(par[4000:] == -1)
        _check_parents(lmc, par, s, e, w, 0)
        # another root: the island is reachable from inside
        lmc_b, par_b, _ = ctx.graph_cost_to_root(4500)
        want_b, _ = _oracle_solve(oracle, n, s, e, w, 4500)
        assert np.array_equal(lmc_b, want_b) and np.all(np.isinf(lmc_b[:4000]))


def test_cost_to_root_edges_of_the_domain():
    with Context(3) as ctx:
        with pytest.raises(_capi.RrtxError):
            ctx.graph_cost_to_root(0)                    # empty tree
        ctx.nodes_append(np.array([[0.0, 0, 0], [3.0, 4.0, 0], [3.0, 4.0, 12.0]]))
        lmc, par, passes = ctx.graph_cost_to_root(1)     # no edges at all
        assert lmc.tolist() == [INF, 0.0, INF] and par.tolist() == [-1, -1, -1] and passes == 0
        ctx.graph_edges_append([1, 2, 0], [0, 1, 0])     # 1 -> 0 (5), 2 -> 1 (12), and a self loop at the root
        lmc, par, _ = ctx.graph_cost_to_root(0)
        assert lmc.tolist() == [0.0, 5.0, 17.0] and par.tolist() == [-1, 0, 1]
        ctx.graph_edges_set_dist(1, [float("nan")])      # a NaN cost never relaxes
        lmc, par, _ = ctx.graph_cost_to_root(0)
        assert lmc.tolist() == [0.0, 5.0, INF] and par.tolist() == [-1, 0, -1]
        for bad in (-1, 3):
            with pytest.raises(_capi.RrtxError):
                ctx.graph_cost_to_root(bad)
        with pytest.raises(_capi.RrtxError):
            ctx.graph_edges_set_dist(2, [1.0, 2.0])      # past the last edge
        with pytest.raises(_capi.RrtxError):
            ctx.graph_edges_block([3])


def test_update_follows_a_replanning_sequence(oracle):
    """rrtx_graph_cost_update through what a planner does between solves: the tree grows (nodes + edges appended),
    obstacles appear (sweep + block), costs are re-priced (set_dist up and down).  After every step the update must
    equal the oracle's reduceInconsistency on the same graph, and a full solve in a second context."""
    rng = np.random.default_rng(11)
    n_total, r, root = 12_000, 3.2, 5
    pts = rng.uniform(-20, 20, (n_total, 3))
    s_all, e_all = _geometric_graph(oracle, pts, r)
    w_all = _edge_dist(pts, s_all, e_all)
    # the tree grows in index order: an edge exists once both its nodes do
    born = np.maximum(s_all, e_all)
    order = np.argsort(born, kind="stable")
    s_all, e_all, w_all, born = s_all[order], e_all[order], w_all[order], born[order]
    stages = [6000, 6001, 6500, 9000, 12_000]
    with Context(3) as ctx, Context(3) as fresh:
        have_n = have_e = 0
        w_now = np.zeros(0)
        for step, n in enumerate(stages):
            ne = int(np.searchsorted(born, n))                 # edges among the first n nodes
            ctx.nodes_append(pts[have_n:n])
            ctx.graph_edges_append(s_all[have_e:ne], e_all[have_e:ne])
            w_now = np.concatenate([w_now, w_all[have_e:ne]])
            have_n, have_e = n, ne
            if step == 2:
                # an obstacle appears next to the root: blocked edges orphan a region
                sph = np.array([[pts[root, 0] + 2.5, pts[root, 1], pts[root, 2], 3.0]])
                ctx.spheres_set(sph, np.ones(1, dtype=np.uint8))
                ids = ctx.obstacle_sweep(0, RR + r + 3.0, RR)
                assert len(ids) > 10
                ctx.graph_edges_block(ids)
                w_now[ids] = INF
            if step == 3:
                # re-pricing: some edges get cheaper, some dearer, some blocked, some unblocked
                pick = rng.choice(ne, 4000, replace=False)
                new = w_now[pick].copy()
                new[:1500] *= 0.25
                new[1500:3000] *= 3.0
                new[3000:3500] = INF
                new[3500:] = w_all[pick[3500:]]
                w_now[pick] = new
                runs = np.split(np.sort(pick), np.nonzero(np.diff(np.sort(pick)) != 1)[0] + 1)
                for run in runs:
                    ctx.graph_edges_set_dist(int(run[0]), w_now[run])
            lmc, par, passes = ctx.graph_cost_update(root)
            want, _ = _oracle_solve(oracle, n, s_all[:ne], e_all[:ne], w_now, root)
            assert np.array_equal(lmc, want), step
            _check_parents(lmc, par, s_all[:ne], e_all[:ne], w_now, root)
            if step > 0:
                assert passes < 60
        # the same graph solved from nothing
        fresh.nodes_append(pts)
        fresh.graph_edges_append(s_all, e_all)
        fresh.graph_edges_set_dist(0, w_now)
        lmc_f, par_f, _ = fresh.graph_cost_to_root(root)
        assert np.array_equal(lmc_f, lmc) and np.array_equal(par_f, par)
        # update with nothing new is the identity; a different root solves in full
        lmc_i, par_i, p_i = ctx.graph_cost_update(root)
        assert np.array_equal(lmc_i, lmc) and np.array_equal(par_i, par) and p_i <= 9
        lmc_o, _, _ = ctx.graph_cost_update(77)
        want_o, _ = _oracle_solve(oracle, n_total, s_all, e_all, w_now, 77)
        assert np.array_equal(lmc_o, want_o)


def test_update_after_blocking_matches_block_propagate_reduce(oracle):
    # the reference's own sequence on the oracle: blockEdge per edge, propogateDescendants, reduceInconsistency
    rng = np.random.default_rng(21)
    n, r, root = 8000, 3.6, 0
    pts = rng.uniform(-20, 20, (n, 3))
    s, e = _geometric_graph(oracle, pts, r)
    w = _edge_dist(pts, s, e)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.graph_edges_append(s, e)
        lmc0, par0, _ = ctx.graph_cost_to_root(root)
        # block the parent edges of 200 nodes (every one orphans a subtree) and 300 other edges
        victims = rng.choice(np.nonzero(par0 >= 0)[0], 200, replace=False)
        blocked = np.unique(np.concatenate([par0[victims], rng.choice(len(s), 300, replace=False)])).astype(np.int32)
        ctx.graph_edges_block(blocked)
        lmc1, par1, passes = ctx.graph_cost_update(root)
        want, _ = _oracle_solve(oracle, n, s, e, w, root, blocked)
        assert np.array_equal(lmc1, want)
        assert np.all(lmc1 >= lmc0) and (lmc1 != lmc0).sum() > 200
        w1 = w.copy()
        w1[blocked] = INF
        _check_parents(lmc1, par1, s, e, w1, root)
        # unblock everything again: back to the first answer
        ctx.graph_edges_set_dist(0, w)
        lmc2, par2, _ = ctx.graph_cost_update(root)
        assert np.array_equal(lmc2, lmc0) and np.array_equal(par2, par0)


def test_update_with_free_edges_falls_back_to_a_full_solve(oracle):
    # parent edges of cost 0 need not form a forest; the update then solves in full and still agrees
    rng = np.random.default_rng(4)
    n = 600
    pts = rng.uniform(-5, 5, (n, 3))
    pts[100:110] = pts[100]                                   # ten nodes on one spot: cost-0 edges among them
    s, e = _geometric_graph(oracle, pts, 2.0)
    w = _edge_dist(pts, s, e)
    assert (w == 0).sum() >= 90
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.graph_edges_append(s, e)
        lmc0, par0, _ = ctx.graph_cost_to_root(3)
        blocked = np.unique(par0[100:110][par0[100:110] >= 0]).astype(np.int32)
        ctx.graph_edges_block(blocked)
        lmc1, _, _ = ctx.graph_cost_update(3)
        w1 = w.copy()
        w1[blocked] = INF
        want, _ = _oracle_solve(oracle, n, s, e, w1, 3)
        assert np.array_equal(lmc1, want)
