"""Node-range sharding of the tree over several contexts (one per GPU in production, two on one GPU here;
SURVEY 8e): each context holds an index range, every context searches all queries, concatenating the
shards' lists per query in shard order gives the unsharded CSR.  Only the context that holds node 0 applies
the root's <= rule (RRTX_OPT_ROOT_RULE)."""
import math

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu


def _merge(parts, bases, nq):
    off = np.zeros(nq + 1, dtype=np.int64)
    idx, dist = [], []
    for i in range(nq):
        for (o, ix, d), b in zip(parts, bases):
            idx.append(ix[o[i]:o[i + 1]] + b)
            dist.append(d[o[i]:o[i + 1]])
        off[i + 1] = off[i] + sum(len(x) for x in idx[-len(parts):])
    return off, np.concatenate(idx), np.concatenate(dist)


@pytest.mark.parametrize("dim", [3, 4])
def test_two_shards_equal_the_whole_tree(dim):
    n, nq, r = 30_000, 600, 4.0 if dim == 3 else 9.0
    pts, Q = synth.nodes(n, dim), synth.queries(nq, dim)
    lo = 17_123
    # exactly at the range from the root (taken, <=) and exactly at the range from shard 1's first node
    # (an ordinary node there: NOT taken, <)
    Q[3] = pts[0]; Q[3, 0] += r
    Q[4] = pts[lo]; Q[4, 0] += r
    d3 = math.sqrt(sum((Q[3, k] - pts[0, k]) * (Q[3, k] - pts[0, k]) for k in range(dim)))
    d4 = math.sqrt(sum((Q[4, k] - pts[lo, k]) * (Q[4, k] - pts[lo, k]) for k in range(dim)))
    if d3 != r or d4 != r:
        pytest.skip("the constructed distances do not round to exactly r on this input")
    with Context(dim) as whole, Context(dim) as a, Context(dim) as b:
        for c in (whole, a, b):
            if dim == 4:
                c.set_wrap(3, 2.0 * math.pi)
        whole.nodes_append(pts)
        a.nodes_append(pts[:lo])
        b.nodes_append(pts[lo:])
        b.set_option(_capi.RRTX_OPT_ROOT_RULE, 0)
        off, idx, dist = whole.nn_radius(Q, r)
        m_off, m_idx, m_dist = _merge([a.nn_radius(Q, r), b.nn_radius(Q, r)], [0, lo], nq)
        assert np.array_equal(off, m_off) and np.array_equal(idx, m_idx) and np.array_equal(dist, m_dist)
        assert 0 in idx[off[3]:off[4]] and lo not in idx[off[4]:off[5]]
        # without the option shard 1 would take its first node with <=
        b.set_option(_capi.RRTX_OPT_ROOT_RULE, 1)
        o2, i2, _ = b.nn_radius(Q[4:5], r)
        assert 0 in i2
        # nearest: lexicographic minimum of the shards' (distance, index)
        ni, nd = whole.nn_nearest(Q)
        ai, ad = a.nn_nearest(Q)
        bi, bd = b.nn_nearest(Q)
        take_b = (bd < ad)
        assert np.array_equal(np.where(take_b, bi + lo, ai), ni) and np.array_equal(np.where(take_b, bd, ad), nd)
