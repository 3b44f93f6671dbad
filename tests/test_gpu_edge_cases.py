"""Empty / ragged / degenerate inputs and the error behaviour of the C-ABI on the GPU."""
import ctypes as C

import numpy as np
import pytest

from rrtqx_3d_amd import _capi
from rrtqx_3d_amd._capi import RrtxError
from rrtqx_3d_amd.context import Context

pytestmark = pytest.mark.gpu


def test_empty_tree_is_a_state_error():
    with Context(3) as ctx:
        with pytest.raises(_capi.RrtxError) as e:
            ctx.nn_radius([[0, 0, 0]], 1.0)
        assert e.value.code == _capi.RRTX_E_STATE and "empty tree" in str(e.value)
        with pytest.raises(_capi.RrtxError):
            ctx.nn_nearest([[0, 0, 0]])


def test_zero_sized_batches():
    with Context(3) as ctx:
        ctx.nodes_append(np.zeros((0, 3)))
        assert ctx.n_nodes == 0
        ctx.nodes_append([[1.0, 2.0, 3.0]])
        off, idx, dist = ctx.nn_radius(np.zeros((0, 3)), 1.0)
        assert list(off) == [0] and len(idx) == 0
        hit, first = ctx.edges_check(np.zeros((0, 3)), np.zeros((0, 3)), 0.5)
        assert len(hit) == 0
        unsafe, clr = ctx.points_check(np.zeros((0, 3)), 0.5)
        assert len(unsafe) == 0
        out = ctx.extend_candidates(np.zeros((0, 3)), 1.0, 0.5)
        assert len(out["idx"]) == 0


def test_no_obstacles_and_all_inactive():
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0], [1, 0, 0]])
        p0, p1 = np.array([[0.0, 0, 0], [5, 5, 5]]), np.array([[1.0, 0, 0], [5, 5, 5]])
        hit, first = ctx.edges_check(p0, p1, 0.5)             # empty obstacle list
        assert list(hit) == [0, 0] and list(first) == [-1, -1]
        unsafe, clr = ctx.points_check(p0, 0.5)
        assert list(unsafe) == [0, 0] and np.isinf(clr).all()   # (false, Inf), R/DRRT_Q.jl:1538,1555
        ctx.spheres_set([[0, 0, 0, 100.0]], active=[0])
        hit, _ = ctx.edges_check(p0, p1, 0.5)                 # zero-length edge vs inactive sphere: no hit (K6)
        assert list(hit) == [0, 0]
        ctx.spheres_set([[0, 0, 0, 100.0]])
        hit, _ = ctx.edges_check(p0, p1, 0.5)
        assert list(hit) == [1, 1]
        out = ctx.extend_candidates([[0.5, 0, 0]], 10.0, 0.5)
        assert list(out["idx"]) == [0, 1] and list(out["hit_out"]) == [1, 1] and out["sample_unsafe"][0] == 1


def test_single_node_tree_and_root_rule():
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        idx, dist = ctx.nn_nearest([[3, 4, 0], [0, 0, 0]])
        assert list(idx) == [0, 0] and list(dist) == [5.0, 0.0]
        off, idx, dist = ctx.nn_radius([[3, 4, 0]], 5.0)      # root: 5.0 <= 5.0
        assert list(idx) == [0] and list(dist) == [5.0]
        off, idx, dist = ctx.nn_radius([[3, 4, 0]], np.nextafter(5.0, 0))
        assert len(idx) == 0
        out = ctx.extend_candidates([[100.0, 0, 0]], 1.0, 0.5)  # empty ball: nearest falls back to the full scan
        assert len(out["idx"]) == 0 and out["nearest_idx"][0] == 0 and out["nearest_dist"][0] == 100.0


def test_incremental_append_matches_bulk(oracle):
    """kdInsert one node at a time (the planner's pattern, incl. capacity growth) == bulk load"""
    rng = np.random.default_rng(8)
    pts = rng.uniform(-10, 10, (3000, 3))
    Q = rng.uniform(-10, 10, (64, 3))
    with Context(3, node_capacity=16) as a, Context(3) as b:
        for k in range(0, 3000, 7):
            first = a.nodes_append(pts[k:k + 7])
            assert first == k
        b.nodes_append(pts)
        ra, rb = a.nn_radius(Q, 2.5), b.nn_radius(Q, 2.5)
        assert all(np.array_equal(x, y) for x, y in zip(ra, rb))
        assert len(ra[1]) > 0


def test_invalid_arguments_return_codes(hip_lib):
    h = C.c_void_p()
    assert hip_lib.rrtx_create(C.byref(h), 5, 0, 16) == _capi.RRTX_E_INVALID          # dim must be 3 or 4
    assert hip_lib.rrtx_create(C.byref(h), 3, 99, 16) == _capi.RRTX_E_INVALID         # device ordinal
    assert hip_lib.rrtx_create(None, 3, 0, 16) == _capi.RRTX_E_INVALID
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0]])
        with pytest.raises(_capi.RrtxError) as e:
            ctx.set_wrap(7, 1.0)
        assert e.value.code == _capi.RRTX_E_INVALID
        with pytest.raises(_capi.RrtxError):
            ctx.obstacle_update(0, 1.0, True)                  # no spheres set
        with pytest.raises(_capi.RrtxError):
            ctx.edges_check_idx([0], [5], 0.5)                 # node index out of range
        with pytest.raises(_capi.RrtxError) as e:
            ctx.dubins_steer(np.zeros((1, 4)), np.zeros((1, 4)), 1.0)   # needs a dim=4 ctx
        assert e.value.code == _capi.RRTX_E_STATE
        assert hip_lib.rrtx_nn_radius(ctx.handle, None, None, 0, 1, None, None, None, 0, None) == _capi.RRTX_E_INVALID
        assert b"bad arguments" in hip_lib.rrtx_last_error(ctx.handle)


def test_nan_inputs_flow_through():
    """NaN is not an error: a NaN query finds nothing, a NaN edge collides with every active sphere"""
    with Context(3) as ctx:
        ctx.nodes_append([[0, 0, 0], [1, 1, 1]])
        ctx.spheres_set([[50, 50, 50, 1.0]])
        off, idx, _ = ctx.nn_radius([[np.nan, 0, 0]], 10.0)
        assert len(idx) == 0
        hit, first = ctx.edges_check([[np.nan, 0, 0]], [[1, 0, 0]], 0.5)
        assert list(hit) == [1] and list(first) == [0]


def test_results_are_deterministic_across_runs():
    """hits are appended with atomics in arbitrary order, then ordered by node index: 10 repeated
    calls must return identical bytes (also exercises the LDS queues / record buffer for races)"""
    from rrtqx_3d_amd import synth
    pts, Q, sph = synth.nodes(50_000, 3), synth.queries(2048, 3), synth.spheres(64)
    r = synth.ball_radius(50_000, 3)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph)
        ref = ctx.extend_candidates(Q, r, 0.5)
        refn = ctx.nn_nearest(Q)
        for _ in range(10):
            out = ctx.extend_candidates(Q, r, 0.5)
            for k in ref:
                assert np.array_equal(out[k], ref[k]), k
            n = ctx.nn_nearest(Q)
            assert np.array_equal(n[0], refn[0]) and np.array_equal(n[1], refn[1])


def test_host_path_registered_and_pageable_outputs_agree():
    """rrtx_extend_candidates moves results through the context's pinned arena, or by direct DMA into arrays the caller
    registered (rrtx_host_register): same bytes either way, also when the arrays are reused across calls of different
    sizes and when capacity runs out (two-call pattern)."""
    from rrtqx_3d_amd import synth
    pts, sph = synth.nodes(20_000, 3), synth.spheres(24)
    with Context(3) as ctx:
        ctx.nodes_append(pts)
        ctx.spheres_set(sph)
        ref = ctx.extend_candidates(synth.queries(700, 3), 5.0, 0.5)
        plain = ctx.extend_out_buffers(700, 40_000)
        pinned = ctx.extend_out_buffers(700, 40_000, register=True)
        for ob in (plain, pinned, plain, pinned):
            got = ctx.extend_candidates(synth.queries(700, 3), 5.0, 0.5, out=ob)
            for k in ref:
                assert np.array_equal(got[k], ref[k]), k
        small = ctx.extend_candidates(synth.queries(300, 3, seed=5), 5.0, 0.5)          # fewer samples through the same arrays
        sub = {k: (v[:301] if k == "offsets" else (v[:300] if v.shape[0] == 700 else v)) for k, v in pinned.items()}
        got = ctx.extend_candidates(synth.queries(300, 3, seed=5), 5.0, 0.5, out=sub)
        for k in small:
            assert np.array_equal(got[k], small[k]), k
        tiny = ctx.extend_out_buffers(700, 100)
        with pytest.raises(RrtxError):
            ctx.extend_candidates(synth.queries(700, 3), 5.0, 0.5, out=tiny)
        for a in pinned.values():
            ctx.host_unregister(a)
        with pytest.raises(RrtxError):
            ctx.host_unregister(pinned["idx"])
