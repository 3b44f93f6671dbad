"""Committed fixtures (tests/golden/hotpath_v2.npz, made by tests/golden/make_golden.py).

CPU leg: the oracle still reproduces them (guards the oracle against drift).
GPU leg: the HIP path reproduces them through the C-ABI without the oracle."""
import json
import math
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(G, "hotpath_v2.npz"), allow_pickle=False)


@pytest.fixture(scope="module")
def env():
    return json.load(open(os.path.join(G, "env_inputs.json")))


def _polys_list_order(env):
    polys = [np.array(p) for p in env["rand_Static_polygons"]][::-1]
    act = (np.array(env["rand_Static_behaviour"]) != 1).astype(np.uint8)[::-1]
    return polys, act


# ------------------------------------------------------------------ CPU ------
def test_oracle_reproduces_sphere_slice(oracle, gold):
    t = oracle.KDTree(3)
    t.insert_many(gold["s_nodes"])
    r = float(gold["s_radius"])
    off, idx, key = gold["s_offsets"], gold["s_idx"], gold["s_key"]
    for i, q in enumerate(gold["s_queries"]):
        a, k = t.within_range(r, q)
        o = np.argsort(a)
        assert np.array_equal(a[o], idx[off[i]:off[i + 1]]) and np.array_equal(k[o], key[off[i]:off[i + 1]])
        ni, nd = t.nearest(q)
        assert ni == gold["s_nearest_idx"][i] and nd == gold["s_nearest_dist"][i]
    sp, m = oracle.make_spheres(gold["s_spheres"], gold["s_active"])
    from rrtqx_3d_amd import synth
    p0, p1 = synth.candidate_edges(gold["s_queries"], gold["s_nodes"], off, idx)
    hit, first = oracle.edges_check_spheres(sp, m, p0, p1, 0.5)
    assert np.array_equal(hit, gold["s_hit"]) and np.array_equal(first, gold["s_first"])
    assert 0 < hit.sum() < len(hit)


def test_oracle_reproduces_polygon_and_dubins_slices(oracle, gold, env):
    polys, act = _polys_list_order(env)
    ps = oracle.PolygonSet(polys, active=act)
    assert np.array_equal(ps.centre_radius(), gold["p_centre_radius"])
    hit, first = oracle.edges_check_polygons(ps, gold["p_edges0"], gold["p_edges1"], 0.5)
    assert np.array_equal(hit, gold["p_hit"]) and np.array_equal(first, gold["p_first"])
    for i in range(0, 256, 7):
        c, w, traj = oracle.dubins_steer(gold["d_start"][i], gold["d_goal"][i], 1.0)
        assert c == gold["d_cost"][i] and w.encode() == gold["d_word"][i] and traj.shape[0] == gold["d_traj_len"][i]


# ------------------------------------------------------------------ GPU ------
@pytest.mark.gpu
def test_gpu_reproduces_sphere_slice(gold):
    from rrtqx_3d_amd.context import Context
    with Context(3) as ctx:
        ctx.nodes_append(gold["s_nodes"])
        ctx.spheres_set(gold["s_spheres"], gold["s_active"])
        out = ctx.extend_candidates(gold["s_queries"], float(gold["s_radius"]), 0.5)
        assert np.array_equal(out["offsets"], gold["s_offsets"])
        assert np.array_equal(out["idx"], gold["s_idx"])
        assert np.array_equal(out["cost"], gold["s_key"])            # bit-exact stored distances
        n = len(gold["s_idx"])
        assert np.array_equal(out["hit_out"], gold["s_hit"][:n])
        assert np.array_equal(out["hit_in"], gold["s_hit"][n:])
        assert np.array_equal(out["nearest_idx"], gold["s_nearest_idx"])
        assert np.array_equal(out["nearest_dist"], gold["s_nearest_dist"])
        assert np.array_equal(out["sample_unsafe"], gold["s_unsafe"])
        unsafe, clr = ctx.points_check(gold["s_queries"], 0.5, quick=True)
        assert np.array_equal(unsafe, gold["s_unsafe"]) and np.array_equal(clr, gold["s_clearance"])
        from rrtqx_3d_amd import synth
        p0, p1 = synth.candidate_edges(gold["s_queries"], gold["s_nodes"], gold["s_offsets"], gold["s_idx"])
        hit, first = ctx.edges_check(p0, p1, 0.5)
        assert np.array_equal(hit, gold["s_hit"]) and np.array_equal(first, gold["s_first"])


@pytest.mark.gpu
def test_gpu_reproduces_polygon_dubins_wrapped_slices(gold, env):
    from rrtqx_3d_amd.context import Context
    polys, act = _polys_list_order(env)
    with Context(4) as ctx:
        ctx.set_wrap(3, 2 * math.pi)
        ctx.nodes_append(gold["w_nodes"])
        ctx.polygons_set(polys, active=act)
        hit, first = ctx.edges_check(gold["p_edges0"], gold["p_edges1"], 0.5, kind=1)
        assert np.array_equal(hit, gold["p_hit"]) and np.array_equal(first, gold["p_first"])
        unsafe, clr = ctx.points_check(gold["p_edges0"], 0.5, kind=1)
        assert np.array_equal(unsafe, gold["p_unsafe"]) and np.array_equal(clr, gold["p_clearance"])
        cost, word, dh, tl = ctx.dubins_edges_check(gold["d_start"], gold["d_goal"], 1.0, 0.5)
        # Dubins: exact (device and oracle share include/rrtx_detmath.h)
        assert np.array_equal(cost, gold["d_cost"]) and np.array_equal(word, gold["d_word"])
        assert np.array_equal(dh, gold["d_hit"]) and np.array_equal(tl, gold["d_traj_len"])
        for op, x, y, ref in ((0, gold["dm_ang"], None, gold["dm_sin"]), (1, gold["dm_ang"], None, gold["dm_cos"]),
                              (2, gold["dm_x"], gold["dm_y"], gold["dm_atan2"]), (3, gold["dm_acos_in"], None, gold["dm_acos"])):
            assert np.array_equal(ctx.detmath_eval(op, x, y).view(np.uint64), ref.view(np.uint64)), op
        off, idx, key = ctx.nn_radius(gold["w_queries"], float(gold["w_radius"]))
        assert np.array_equal(off, gold["w_offsets"]) and np.array_equal(idx, gold["w_idx"])
        assert np.array_equal(key, gold["w_key"])
