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
            near = d_ni.cpu().numpy()
            has = np.diff(out["offsets"]) > 0
            assert np.array_equal(near[has], out["nearest_idx"][has])      # empty balls: -1 here, resolved by the host form
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
