"""The device-pointer entry points (*_dev: raw HIP pointers, caller's stream, no host sync) must
give what the host-pointer forms give (those are checked against the oracle elsewhere)."""
import numpy as np
import pytest

from rrtqx_3d_amd import synth
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu
RR = 0.5


def test_dev_entry_points_match_host_forms():
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, nq = 30_000, 700
    pts, Q, sph = synth.nodes(n, 3), synth.queries(nq, 3), synth.spheres(48)
    r = synth.ball_radius(n, 3)
    with Context(3) as host, Context(3) as dctx:
        host.nodes_append(pts)
        host.spheres_set(sph)
        dctx.spheres_set(sph)
        st = torch.cuda.Stream(device=dev)
        dctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            d_pts = torch.from_numpy(pts).to(dev)
            d_q = torch.from_numpy(Q).to(dev)
            st.synchronize()
            dctx.nodes_append_dev(d_pts.data_ptr(), 10_000)              # in two pieces
            dctx.nodes_append_dev(d_pts.data_ptr() + 10_000 * 24, n - 10_000)
            assert dctx.n_nodes == n
            # nearest
            d_ni = torch.empty(nq, dtype=torch.int32, device=dev)
            d_nd = torch.empty(nq, dtype=torch.float64, device=dev)
            dctx.nn_nearest_dev(d_q.data_ptr(), nq, d_ni.data_ptr(), d_nd.data_ptr())
            hi, hd = host.nn_nearest(Q)
            st.synchronize()
            assert np.array_equal(d_ni.cpu().numpy(), hi) and np.array_equal(d_nd.cpu().numpy(), hd)
            # radius (capacity known from the host form)
            off, idx, dist = host.nn_radius(Q, r)
            cap = len(idx) + 7
            d_off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
            d_idx = torch.empty(cap, dtype=torch.int32, device=dev)
            d_dist = torch.empty(cap, dtype=torch.float64, device=dev)
            d_need = torch.zeros(1, dtype=torch.int64, device=dev)
            dctx.nn_radius_dev(d_q.data_ptr(), r, nq, d_off.data_ptr(), d_idx.data_ptr(), d_dist.data_ptr(), cap,
                               d_need.data_ptr())
            st.synchronize()
            k = int(d_need.item())
            assert k == len(idx) and np.array_equal(d_off.cpu().numpy(), off)
            assert np.array_equal(d_idx.cpu().numpy()[:k], idx) and np.array_equal(d_dist.cpu().numpy()[:k], dist)
            # too small a capacity: the count is still exact, nothing is written past the capacity
            d_small = torch.full((16,), -7, dtype=torch.int32, device=dev)
            d_sd = torch.empty(16, dtype=torch.float64, device=dev)
            dctx.nn_radius_dev(d_q.data_ptr(), r, nq, d_off.data_ptr(), d_small.data_ptr(), d_sd.data_ptr(), 8,
                               d_need.data_ptr())
            st.synchronize()
            assert int(d_need.item()) == len(idx) and (d_small.cpu().numpy()[8:] == -7).all()
            # fused extend preamble
            out = host.extend_candidates(Q, r, RR)
            d_cost = torch.empty(cap, dtype=torch.float64, device=dev)
            d_ho = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_hi = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_un = torch.empty(nq, dtype=torch.uint8, device=dev)
            dctx.extend_candidates_dev(d_q.data_ptr(), nq, r, RR, d_off.data_ptr(), d_idx.data_ptr(), d_cost.data_ptr(),
                                       d_ho.data_ptr(), d_hi.data_ptr(), cap, d_need.data_ptr(), d_ni.data_ptr(),
                                       d_nd.data_ptr(), d_un.data_ptr())
            st.synchronize()
            k = int(d_need.item())
            assert k == len(out["idx"])
            assert np.array_equal(d_idx.cpu().numpy()[:k], out["idx"]) and np.array_equal(d_cost.cpu().numpy()[:k], out["cost"])
            assert np.array_equal(d_ho.cpu().numpy()[:k], out["hit_out"]) and np.array_equal(d_hi.cpu().numpy()[:k], out["hit_in"])
            assert np.array_equal(d_un.cpu().numpy(), out["sample_unsafe"])
            # empty balls are answered on the device too (expanding search): no -1 leaves the library
            assert np.array_equal(d_ni.cpu().numpy(), out["nearest_idx"])
            assert np.array_equal(d_nd.cpu().numpy(), out["nearest_dist"])
            # stand-alone edge and point checks
            p0, p1 = synth.candidate_edges(Q, pts, out["offsets"], out["idx"])
            ne = len(p0)
            d_p0, d_p1 = torch.from_numpy(p0).to(dev), torch.from_numpy(p1).to(dev)
            d_hit = torch.empty(ne, dtype=torch.uint8, device=dev)
            d_first = torch.empty(ne, dtype=torch.int32, device=dev)
            st.synchronize()
            dctx.edges_check_dev(0, d_p0.data_ptr(), d_p1.data_ptr(), ne, RR, -1, -1, -1, d_hit.data_ptr(), d_first.data_ptr())
            hh, hf = host.edges_check(p0, p1, RR)
            st.synchronize()
            assert np.array_equal(d_hit.cpu().numpy(), hh) and np.array_equal(d_first.cpu().numpy(), hf)
            d_clr = torch.empty(nq, dtype=torch.float64, device=dev)
            dctx.points_check_dev(0, d_q.data_ptr(), nq, RR, True, d_un.data_ptr(), d_clr.data_ptr())
            hu, hc = host.points_check(Q, RR)
            st.synchronize()
            assert np.array_equal(d_un.cpu().numpy(), hu) and np.array_equal(d_clr.cpu().numpy(), hc)
        dctx.set_stream(None)


def test_dubins_preamble_dev_matches_host_form():
    torch = pytest.importorskip("torch")
    import math
    dev = torch.device("cuda", 0)
    n, nq, r, r_min = 12_000, 300, 9.0, 1.0
    pts, Q, polys = synth.nodes(n, 4), synth.queries(nq, 4), synth.polygons(24)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2 * math.pi)
        ctx.nodes_append(pts)
        ctx.polygons_set(polys)
        ref = ctx.extend_candidates_dubins(Q, r, RR, r_min)
        k = len(ref["idx"])
        cap = k + 11
        st = torch.cuda.Stream(device=dev)
        ctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            d_q = torch.from_numpy(Q).to(dev)
            f64 = lambda m: torch.empty(m, dtype=torch.float64, device=dev)
            u8 = lambda m: torch.empty(m, dtype=torch.uint8, device=dev)
            d_off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
            d_idx = torch.empty(cap, dtype=torch.int32, device=dev)
            d_key, d_co, d_ci = f64(cap), f64(cap), f64(cap)
            d_wo, d_wi, d_ho, d_hi, d_un = u8(3 * cap), u8(3 * cap), u8(cap), u8(cap), u8(nq)
            d_need = torch.zeros(1, dtype=torch.int64, device=dev)
            d_ni = torch.empty(nq, dtype=torch.int32, device=dev)
            d_nd = f64(nq)
            st.synchronize()
            ctx.extend_candidates_dubins_dev(d_q.data_ptr(), nq, r, RR, r_min, d_off.data_ptr(), d_idx.data_ptr(),
                                             d_key.data_ptr(), d_co.data_ptr(), d_ci.data_ptr(), d_wo.data_ptr(),
                                             d_wi.data_ptr(), d_ho.data_ptr(), d_hi.data_ptr(), cap, d_need.data_ptr(),
                                             d_ni.data_ptr(), d_nd.data_ptr(), d_un.data_ptr())
            st.synchronize()
        ctx.set_stream(None)
        assert int(d_need.item()) == k and k > 1000
        assert np.array_equal(d_off.cpu().numpy(), ref["offsets"])
        for name, t in (("idx", d_idx), ("key", d_key), ("cost_out", d_co), ("cost_in", d_ci), ("hit_out", d_ho), ("hit_in", d_hi)):
            assert np.array_equal(t.cpu().numpy()[:k], ref[name]), name
        assert np.array_equal(d_un.cpu().numpy(), ref["sample_unsafe"])
        assert np.array_equal(d_ni.cpu().numpy(), ref["nearest_idx"]) and np.array_equal(d_nd.cpu().numpy(), ref["nearest_dist"])


def test_knearest_dev_matches_host_form():
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, nq, k = 40_000, 900, 9
    pts, Q = synth.nodes(n, 3), synth.queries(nq, 3)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ref = ctx.nn_knearest(Q, k)
        st = torch.cuda.Stream(device=dev)
        ctx.set_stream(st.cuda_stream)
        with torch.cuda.stream(st):
            d_q = torch.from_numpy(Q).to(dev)
            d_idx = torch.empty((nq, k), dtype=torch.int32, device=dev)
            d_dist = torch.empty((nq, k), dtype=torch.float64, device=dev)
            d_cnt = torch.empty(nq, dtype=torch.int32, device=dev)
            st.synchronize()
            ctx.nn_knearest_dev(d_q.data_ptr(), nq, k, d_idx.data_ptr(), d_dist.data_ptr(), d_cnt.data_ptr())
            st.synchronize()
        ctx.set_stream(None)
        assert np.array_equal(d_idx.cpu().numpy(), ref[0]) and np.array_equal(d_dist.cpu().numpy(), ref[1])
        assert np.array_equal(d_cnt.cpu().numpy(), ref[2])


def _naive_nearest(pts, q):
    d = pts - q
    s = d[:, 0] * d[:, 0]
    for k in range(1, pts.shape[1]):
        s = s + d[:, k] * d[:, k]
    i = int(np.argmin(s))            # first minimum = lowest index among ties
    return i, float(np.sqrt(s[i]))


def test_nearest_dev_adversarial_order_small_record_buffer():
    """rrtx_nn_nearest_dev has no host round trip, so an overflowing candidate buffer must be repaired on
    the device: nodes sorted by DEcreasing distance make every node a new running minimum
    (R/kdTree_general.jl:357-385 still has one answer); with the record capacity forced tiny every query
    loses records and is answered by the exact fix-up kernel."""
    torch = pytest.importorskip("torch")
    from rrtqx_3d_amd import _capi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    n = 30_000
    radii = np.linspace(90.0, 1.0, n)
    u = rng.normal(size=(n, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    pts = u * radii[:, None]
    pts[n - 5] = pts[n - 1]                                   # exact tie at the minimum: the lower index wins
    Q = np.concatenate([np.zeros((3, 3)), rng.uniform(-5, 5, (61, 3))])
    nq = len(Q)
    for cap in (0, 64):
        with Context(3) as ctx:
            ctx.nodes_append(pts)
            ctx.set_option(_capi.RRTX_OPT_NEAREST_REC_CAP, cap)
            d_q = torch.from_numpy(Q).to(dev)
            d_ni = torch.full((nq,), -9, dtype=torch.int32, device=dev)
            d_nd = torch.empty(nq, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            ctx.nn_nearest_dev(d_q.data_ptr(), nq, d_ni.data_ptr(), d_nd.data_ptr())
            ctx.sync()
            gi, gd = d_ni.cpu().numpy(), d_nd.cpu().numpy()
            for k in range(nq):
                ri, rd = _naive_nearest(pts, Q[k])
                assert gi[k] == ri and gd[k] == rd, (cap, k)
            assert gi[0] == n - 5


def test_nearest_dev_fixup_with_wrapped_dimension():
    """the same repair with ghosts: [x y t theta], theta wrapped -- equals the exact fp64 scan"""
    torch = pytest.importorskip("torch")
    import math
    from rrtqx_3d_amd import _capi
    dev = torch.device("cuda", 0)
    n, nq = 9_000, 200
    pts, Q = synth.nodes(n, 4), synth.queries(nq, 4)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2 * math.pi)
        ctx.nodes_append(pts)
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
        ei, ed = ctx.nn_nearest(Q)                            # exact scan
        ctx.set_option(_capi.RRTX_OPT_NN_FILTER, 1)
        ctx.set_option(_capi.RRTX_OPT_NEAREST_REC_CAP, 32)
        d_q = torch.from_numpy(Q).to(dev)
        d_ni = torch.empty(nq, dtype=torch.int32, device=dev)
        d_nd = torch.empty(nq, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        ctx.nn_nearest_dev(d_q.data_ptr(), nq, d_ni.data_ptr(), d_nd.data_ptr())
        ctx.sync()
        assert np.array_equal(d_ni.cpu().numpy(), ei) and np.array_equal(d_nd.cpu().numpy(), ed)


def test_extend_dev_resolves_empty_balls_on_device(oracle):
    """samples whose ball holds no node (outside the cloud, or a tiny radius) get kdFindNearest's
    answer from the device-side expanding search; culled and unculled trees, growing tail included"""
    torch = pytest.importorskip("torch")
    from rrtqx_3d_amd import _capi
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    for n, cull in ((3_000, 1), (40_000, 1), (40_000, 2)):
        pts = rng.uniform(-50, 50, (n, 3))
        extra = rng.uniform(-50, 50, (700, 3))                # appended later: sits in the index tail
        Q = np.concatenate([rng.uniform(-50, 50, (300, 3)), rng.uniform(-400, 400, (300, 3)),
                            np.array([[1e6, -1e6, 3.0], [50.0, 50.0, 50.0], [np.nan, 0.0, 0.0]])])
        nq = len(Q)
        r = 0.9                                               # most balls are empty
        allp = np.concatenate([pts, extra])
        with Context(3) as ctx:
            ctx.set_option(_capi.RRTX_OPT_NN_CULL, cull)
            ctx.nodes_append(pts)
            ctx.spheres_set(synth.spheres(16))
            d_q = torch.from_numpy(Q).to(dev)
            cap = 64 * nq
            d_off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
            d_idx = torch.empty(cap, dtype=torch.int32, device=dev)
            d_cost = torch.empty(cap, dtype=torch.float64, device=dev)
            d_ho = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_hi = torch.empty(cap, dtype=torch.uint8, device=dev)
            d_un = torch.empty(nq, dtype=torch.uint8, device=dev)
            d_need = torch.zeros(1, dtype=torch.int64, device=dev)
            d_ni = torch.empty(nq, dtype=torch.int32, device=dev)
            d_nd = torch.empty(nq, dtype=torch.float64, device=dev)
            torch.cuda.synchronize()
            for step, cloud in enumerate((pts, allp)):
                if step == 1:
                    ctx.nodes_append(extra)
                ctx.extend_candidates_dev(d_q.data_ptr(), nq, r, RR, d_off.data_ptr(), d_idx.data_ptr(), d_cost.data_ptr(),
                                          d_ho.data_ptr(), d_hi.data_ptr(), cap, d_need.data_ptr(), d_ni.data_ptr(),
                                          d_nd.data_ptr(), d_un.data_ptr())
                ctx.sync()
                gi, gd = d_ni.cpu().numpy(), d_nd.cpu().numpy()
                off = d_off.cpu().numpy()
                assert (np.diff(off) == 0).sum() > 300
                for k in range(nq - 1):
                    ri, rd = _naive_nearest(cloud, Q[k])
                    assert gi[k] == ri and gd[k] == rd, (n, cull, step, k)
                assert gi[nq - 1] == 0x7fffffff and np.isinf(gd[nq - 1])      # NaN sample orders against nothing


def test_radius_lists_that_outgrow_their_buckets(oracle):
    """a dense cluster makes a few lists many times longer than the average the capacity allows for:
    their records take the shared overflow list and come back (a) collected by the finish kernel itself,
    then, once the library has seen the overflow, (b) with wider buckets / (c) pre-scattered -- every call
    must give the same lists as the exact brute-force path"""
    from rrtqx_3d_amd import _capi
    rng = np.random.default_rng(3)
    pts = np.concatenate([rng.uniform(-50, 50, (20_000, 3)), rng.normal(0, 0.8, (6_000, 3)) + [10, 10, 10]])
    Q = np.concatenate([rng.uniform(-50, 50, (400, 3)), rng.normal(0, 1.0, (40, 3)) + [10, 10, 10]])
    r = 3.0
    with Context(3) as ref:
        ref.set_option(_capi.RRTX_OPT_NN_FILTER, 0)
        ref.nodes_append(pts)
        roff, ridx, rdist = ref.nn_radius(Q, r)
    assert np.diff(roff).max() > 2500 and np.median(np.diff(roff)) < 10
    tree = oracle.KDTree(3)
    tree.insert_many(pts)
    for k in (5, 411, 425):
        oi, od = tree.within_range(r, Q[k])
        o = np.argsort(oi)
        assert np.array_equal(np.asarray(oi)[o], ridx[roff[k]:roff[k + 1]])
        assert np.array_equal(np.asarray(od)[o], rdist[roff[k]:roff[k + 1]])
    for cull in (2, 0):
        with Context(3) as ctx:
            ctx.set_option(_capi.RRTX_OPT_NN_CULL, cull)
            ctx.nodes_append(pts)
            for call in range(4):                               # tight capacity: buckets of 2 x the average
                off, idx, dist = ctx.nn_radius(Q, r, cap=len(ridx))
                assert np.array_equal(off, roff) and np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (cull, call)
            ctx.spheres_set(synth.spheres(24))
            out = ctx.extend_candidates(Q, r, RR, cap=len(ridx))
            assert np.array_equal(out["idx"], ridx) and np.array_equal(out["cost"], rdist)
            p0, p1 = synth.candidate_edges(Q, pts, roff, ridx)
            hh, _ = ctx.edges_check(p0, p1, RR)
            n = len(ridx)
            assert np.array_equal(out["hit_out"], hh[:n]) and np.array_equal(out["hit_in"], hh[n:])


@pytest.mark.parametrize("moving", [False, True])
def test_extend_dev_polygons_without_the_sample_flags(moving):
    """rrtx_extend_candidates_dev against the polygon list with sample_unsafe == NULL: the sample pass then runs for the
    per-sample obstacle lists alone (obstacles that stand still) or not at all (obstacles that move in time: the edge
    kernel lists for itself) -- the edge flags are those of the call that asks for the sample flags too."""
    torch = pytest.importorskip("torch")
    from rrtqx_3d_amd import _capi
    dev = torch.device("cuda", 0)
    n, nq = 40_000, 1500
    pts, Q = synth.nodes(n, 3), synth.queries(nq, 3)
    if moving:
        polys, kinds, paths, active, _ = synth.dynamic_polygons(128)
    else:
        polys, kinds, paths, active = synth.polygons(128), None, None, None
    r = synth.ball_radius(n, 3)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.polygons_set(polys, kinds=kinds, paths=paths, active=active)
        ctx.set_option(_capi.RRTX_OPT_EXTEND_OBSTACLES, 1)
        d_q = torch.from_numpy(Q).to(dev)
        cap = 96 * nq
        res = []
        for want_flags in (True, False):
            d_off = torch.empty(nq + 1, dtype=torch.int64, device=dev)
            d_idx = torch.empty(cap, dtype=torch.int32, device=dev)
            d_cost = torch.empty(cap, dtype=torch.float64, device=dev)
            d_ho = torch.zeros(cap, dtype=torch.uint8, device=dev)
            d_hi = torch.zeros(cap, dtype=torch.uint8, device=dev)
            d_un = torch.zeros(nq, dtype=torch.uint8, device=dev)
            d_need = torch.zeros(1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            ctx.extend_candidates_dev(d_q.data_ptr(), nq, r, RR, d_off.data_ptr(), d_idx.data_ptr(), d_cost.data_ptr(),
                                      d_ho.data_ptr(), d_hi.data_ptr(), cap, d_need.data_ptr(), None, None,
                                      d_un.data_ptr() if want_flags else None)
            ctx.sync()
            total = int(d_off.cpu().numpy()[-1])
            assert 0 < total <= cap
            res.append((d_off.cpu().numpy(), d_idx.cpu().numpy()[:total], d_ho.cpu().numpy()[:total], d_hi.cpu().numpy()[:total]))
        for a, b in zip(res[0], res[1]):
            assert np.array_equal(a, b)
        assert 0 < res[0][2].sum() < len(res[0][2])
