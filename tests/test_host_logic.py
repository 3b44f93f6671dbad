"""CPU tests of the product's host-side logic (no GPU): squared-distance thresholds, environment
file readers, synthetic workload generators, and the oracle's sanitised self test."""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest

from rrtqx_3d_amd import envio, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _thr(lib, r):
    ge, gt = C.c_double(), C.c_double()
    assert lib.rrtx_sq_thresholds(r, C.byref(ge), C.byref(gt)) == 0
    return ge.value, gt.value


def test_sq_thresholds_are_exact(hip_lib):
    rng = np.random.default_rng(0)
    rs = np.concatenate([rng.uniform(0, 20, 2000), 10.0 ** rng.uniform(-300, 150, 500),
                         [3.1497206024977595, 7.783652738915254, 5.0, 1.0, 2.0, 0.5, 1e-310, 4.9e-324]])
    for r in rs:
        ge, gt = _thr(hip_lib, float(r))
        # sqrt(s) < r  <=>  s < ge
        assert math.sqrt(ge) >= r
        if ge > 0:
            assert math.sqrt(np.nextafter(ge, -np.inf)) < r
        # sqrt(s) <= r  <=>  s < gt
        assert math.sqrt(gt) > r
        assert gt == 0 or math.sqrt(np.nextafter(gt, -np.inf)) <= r
        assert ge <= gt


def test_sq_thresholds_special_values(hip_lib):
    assert _thr(hip_lib, 0.0)[0] == 0.0                      # sqrt(s) < 0 never; s < 0 never
    assert _thr(hip_lib, -1.0) == (0.0, 0.0)
    ge, gt = _thr(hip_lib, float("inf"))
    assert ge == float("inf") and math.isnan(gt)             # sqrt(s) > inf never
    ge, gt = _thr(hip_lib, float("nan"))
    assert math.isnan(ge) and gt == 0.0
    # sqrt compresses: several s round to sqrt(s) == 5.0, so the thresholds straddle 25 by a few ulps
    ge, gt = _thr(hip_lib, 5.0)
    assert ge < 25.0 < gt and math.sqrt(ge) == 5.0 and math.sqrt(np.nextafter(gt, 0)) == 5.0
    assert (gt - ge) / 25.0 < 1e-15


def test_env_readers_roundtrip(tmp_path):
    sph = envio.SphereEnv(np.array([[-14.0, -14.0, -18.0, 3.5], [1.5, 2.25, -3.0, 0.78]]), np.array([1, 0]))
    p = tmp_path / "s.txt"
    envio.write_sphere_obstacles(str(p), sph)
    back = envio.read_sphere_obstacles(str(p))
    assert np.array_equal(back.cxyzr, sph.cxyzr) and np.array_equal(back.behaviour, sph.behaviour)
    assert list(back.active()) == [0, 1]                      # behaviour 1 ("appears") starts unused
    c, a = back.list_order()
    assert np.array_equal(c, sph.cxyzr[::-1]) and list(a) == [1, 0]   # listPush: front = last in file
    poly = envio.PolygonEnv([np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]), np.array([[2.0, 2.0], [3.0, 2.0], [3.0, 3.0], [2.0, 3.0]])],
                            np.array([0, -1]))
    q = tmp_path / "p.txt"
    envio.write_polygon_obstacles(str(q), poly)
    pb = envio.read_polygon_obstacles(str(q))
    assert all(np.array_equal(x, y) for x, y in zip(pb.polygons, poly.polygons)) and list(pb.behaviour) == [0, -1]
    bad = tmp_path / "bad.txt"
    bad.write_text("1\n0.0, 0.0, 0.0\n1.0\n7\n")
    with pytest.raises(ValueError, match="unknown behavoiur type"):
        envio.read_sphere_obstacles(str(bad))
    assert list(envio.str2array("1.5, -2, 3e2\n")) == [1.5, -2.0, 300.0]


def test_julia_float_printing_and_tree_dumps(tmp_path):
    """The dump files the MATLAB viewers read (R/DRRT_Q.jl:250-364) are written with Julia's writedlm;
    the kd-free writers of the mirror print numbers the same way (base/grisu/grisu.jl `_show`:
    shortest digits, plain notation for -4 < pt <= 6)."""
    cases = {1.0: "1.0", 0.1: "0.1", 1e-5: "1.0e-5", 1e-4: "0.0001", 123456.789: "123456.789", 999999.0: "999999.0",
             1e6: "1.0e6", 1234567.0: "1.234567e6", 1.5e10: "1.5e10", -2.5: "-2.5", 0.0: "0.0", 100000.0: "100000.0",
             0.00012: "0.00012", 1e21: "1.0e21", 5e-324: "5.0e-324", math.inf: "Inf", -math.inf: "-Inf"}
    for x, want in cases.items():
        assert envio.jl_float_str(x) == want
        if math.isfinite(x):
            assert float(envio.jl_float_str(x)) == x
    assert envio.jl_float_str(math.nan) == "NaN"
    rng = np.random.default_rng(0)
    for x in np.concatenate([rng.uniform(-100, 100, 2000), 10.0 ** rng.uniform(-12, 25, 2000)]):
        assert float(envio.jl_float_str(x)) == x                       # round trip, whatever the notation
    from types import SimpleNamespace as NS
    from rrtqx_3d_amd import drrt
    root = NS(position=np.array([[15.0, 15.0, 15.0]]), rrtTreeCost=0.0, rrtLMC=0.0, rrtParentUsed=False, rrtParentEdge=None)
    kid = NS(position=np.array([[1.5, -2.0, 1e-5]]), rrtTreeCost=21.25, rrtLMC=math.inf, rrtParentUsed=True,
             rrtParentEdge=NS(endNode=root))
    tree = NS(nodes=[root, kid])
    drrt.saveRRTNodes(tree, str(tmp_path / "nodes.txt"))
    drrt.saveRRTTree(tree, str(tmp_path / "edges.txt"))
    drrt.saveRRTNodesCollision(tree, str(tmp_path / "cnodes.txt"))
    assert open(tmp_path / "nodes.txt").read() == "15.0,15.0,15.0,0.0,0.0\n1.5,-2.0,1.0e-5,21.25,Inf\n"
    assert open(tmp_path / "edges.txt").read() == "1.5,-2.0,1.0e-5,21.25\n15.0,15.0,15.0,0.0\n"
    assert open(tmp_path / "cnodes.txt").read() == "15.0,15.0,15.0,0.0\n1.5,-2.0,1.0e-5,21.25\n"


def test_synth_workloads_match_the_survey():
    assert abs(synth.ball_radius(200_000, 3) - 3.1497) < 1e-4 and abs(synth.ball_radius(10_000, 3) - 7.7837) < 1e-4
    assert synth.ball_radius(50_000, 4, gamma=100.0, delta=10.0) == 10.0
    n = synth.nodes(1000, 3)
    assert n.shape == (1000, 3) and np.abs(n).max() <= 50 and np.array_equal(n, synth.nodes(1000, 3))
    d = synth.nodes(100, 4)
    assert (d[:, 2] == 0).all() and (0 <= d[:, 3]).all() and (d[:, 3] < 2 * math.pi).all()
    s = synth.spheres(256)
    assert s.shape == (256, 4) and (1.0 <= s[:, 3]).all() and (s[:, 3] <= 3.5).all()
    p = synth.polygons(64)
    assert len(p) == 64 and all(v.shape[0] in (3, 4) and v.shape[1] == 2 for v in p)
    off = np.array([0, 2, 3]); idx = np.array([5, 7, 1])
    q = np.arange(6.0).reshape(2, 3); pts = np.arange(30.0).reshape(10, 3)
    p0, p1 = synth.candidate_edges(q, pts, off, idx)
    assert p0.shape == (6, 3) and np.array_equal(p0[:3], q[[0, 0, 1]]) and np.array_equal(p1[:3], pts[idx])
    assert np.array_equal(p0[3:], p1[:3]) and np.array_equal(p1[3:], p0[:3])


def test_oracle_sanitizer_selftest():
    """address + undefined-behaviour sanitised run of the oracle (GPU sanitizers are unavailable)"""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "selftest"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout and "runtime error" not in r.stderr
