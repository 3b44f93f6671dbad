"""north_star's obstacle-set shard on the real kernels (one context per GPU in production, three on one GPU
here): every context holds a slice of the obstacle list and checks ALL candidate edges against it; the per-edge flags
combine with a byte-wise maximum (what parallel.reduce_obstacle_shards does with all_reduce(MAX)) and the first-hit
positions with a minimum after adding the slice's base -- equal to the unsharded context's answers, for the sphere list,
the polygon list (fused preamble and stand-alone edge checks) and the Dubins preamble."""
import math

import numpy as np
import pytest

from rrtqx_3d_amd import _capi, parallel, synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
RR = 0.5


def _combine_first(firsts, bases):
    big = np.iinfo(np.int32).max
    g = [np.where(f >= 0, f + b, big) for f, b in zip(firsts, bases)]
    m = np.minimum.reduce(g)
    return np.where(m == big, -1, m).astype(np.int32)


@pytest.mark.parametrize("kind", ["spheres", "polygons"])
def test_fused_preamble_and_edge_checks_over_obstacle_shards(kind):
    n, b, m, shards = 40_000, 1500, 96, 3
    pts, Q = synth.nodes(n, 3), synth.queries(b, 3)
    r = 5.0
    obs = synth.spheres(m) if kind == "spheres" else synth.polygons(m)
    active = np.ones(m, dtype=np.uint8)
    active[[7, 40]] = 0

    def make(lo, hi):
        c = Context(3)
        c.nodes_append(pts)
        if kind == "spheres":
            c.spheres_set(obs[lo:hi], active[lo:hi])
        else:
            c.polygons_set(obs[lo:hi], active=active[lo:hi])
            c.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        return c

    whole = make(0, m)
    spans = [parallel.shard_range(m, k, shards) for k in range(shards)]
    parts = [make(lo, hi) for lo, hi in spans]
    try:
        ref = whole.extend_candidates(Q, r, RR)
        outs = [c.extend_candidates(Q, r, RR) for c in parts]
        for o in outs:                                   # the search does not look at obstacles
            assert np.array_equal(o["offsets"], ref["offsets"]) and np.array_equal(o["idx"], ref["idx"])
            assert np.array_equal(o["cost"], ref["cost"])
        for k in ("hit_out", "hit_in", "sample_unsafe"):
            assert np.array_equal(np.maximum.reduce([o[k] for o in outs]), ref[k]), k
        assert 0 < ref["hit_out"].sum() < len(ref["hit_out"])
        # stand-alone checks with first-hit positions: minimum over the shards of (local position + base)
        p0, p1 = synth.candidate_edges(Q[:300], pts, ref["offsets"][:301], ref["idx"][:ref["offsets"][300]])
        kk = 0 if kind == "spheres" else 1
        hit, first = whole.edges_check(p0, p1, RR, kind=kk)
        hs, fs = zip(*[c.edges_check(p0, p1, RR, kind=kk) for c in parts])
        assert np.array_equal(np.maximum.reduce(hs), hit)
        assert np.array_equal(_combine_first(fs, [lo for lo, _ in spans]), first)
        # the same through the obstacle RANGE arguments of one context that holds the whole list (rrtx_edges_check_dev's
        # obs_begin / obs_end: first hits are then global list positions already)
        import torch
        dev = torch.device("cuda", 0)
        d0, d1 = torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev)
        hs2, fs2 = [], []
        whole.set_stream(torch.cuda.current_stream().cuda_stream)
        for lo, hi in spans:
            dh = torch.zeros(len(p0), dtype=torch.uint8, device=dev)
            df = torch.zeros(len(p0), dtype=torch.int32, device=dev)
            whole.edges_check_dev(kk, d0.data_ptr(), d1.data_ptr(), len(p0), RR, -1, lo, hi, dh.data_ptr(), df.data_ptr())
            whole.sync()
            hs2.append(dh.cpu().numpy()); fs2.append(df.cpu().numpy())
        whole.set_stream(None)
        assert np.array_equal(np.maximum.reduce(hs2), hit)
        assert np.array_equal(_combine_first(fs2, [0] * shards), first)
    finally:
        for c in parts + [whole]:
            c.close()


def test_dubins_preamble_over_obstacle_shards():
    n, b, m = 12_000, 200, 48
    pts, Q, polys = synth.nodes(n, 4), synth.queries(b, 4), synth.polygons(48)
    r, r_min = 6.0, 1.0
    ctxs = []
    try:
        for lo, hi in [(0, m)] + [parallel.shard_range(m, k, 2) for k in range(2)]:
            c = Context(4)
            c.set_wrap(3, 2.0 * math.pi)
            c.nodes_append(pts)
            c.polygons_set(polys[lo:hi])
            ctxs.append(c)
        ref, a, bb = [c.extend_candidates_dubins(Q, r, RR, r_min) for c in ctxs]
        for o in (a, bb):
            assert np.array_equal(o["idx"], ref["idx"]) and np.array_equal(o["cost_out"], ref["cost_out"])
            assert np.array_equal(o["word_in"], ref["word_in"])
        for k in ("hit_out", "hit_in", "sample_unsafe"):
            assert np.array_equal(np.maximum(a[k], bb[k]), ref[k]), k
        assert 0 < ref["hit_out"].sum() < len(ref["hit_out"])
    finally:
        for c in ctxs:
            c.close()
