"""include/rrtx_detmath.h -- the ONE deterministic sin / cos / atan2 / acos compiled into the HIP kernels and
into the oracle (CPU side; the device build is held against the host build in tests/test_gpu_parity.py::
test_detmath_device_equals_host and against the committed bit patterns in tests/test_golden.py).

What is pinned here: (1) this build of the header reproduces the committed bit patterns (compiler / flag drift),
(2) its distance from glibc -- the stand-in for "some other correctly working libm", e.g. Julia's -- is at most
2 ulp, so Dubins costs computed with it stay far inside north_star's 1e-6, (3) the IEEE special cases, (4) the
oracle built with glibc's transcendentals and one cos / sin per arc row (the reference's literal form,
librrtx_oracle_libm.so) agrees with the default build within rounding on poses in general position."""
import ctypes as C
import math
import os

import numpy as np
import pytest

from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ulps(a, b):
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(b), 1e-300))


def test_header_reproduces_committed_bit_patterns():
    g = np.load(os.path.join(G, "hotpath_v2.npz"), allow_pickle=False)
    for op, x, y, ref in ((O.DM_SIN, g["dm_ang"], None, g["dm_sin"]), (O.DM_COS, g["dm_ang"], None, g["dm_cos"]),
                          (O.DM_ATAN2, g["dm_x"], g["dm_y"], g["dm_atan2"]), (O.DM_ACOS, g["dm_acos_in"], None, g["dm_acos"])):
        assert np.array_equal(O.dm_eval(op, x, y).view(np.uint64), ref.view(np.uint64)), op


def test_accuracy_against_libm():
    rng = np.random.default_rng(1)
    x = rng.uniform(-40, 40, 1_000_000)
    assert _ulps(O.dm_eval(O.DM_SIN, x), np.sin(x)).max() <= 1.0
    assert _ulps(O.dm_eval(O.DM_COS, x), np.cos(x)).max() <= 1.0
    x = rng.uniform(-1e5, 1e5, 200_000)                       # 118 bits of pi/2: still exact to the ulp far out
    assert _ulps(O.dm_eval(O.DM_SIN, x), np.sin(x)).max() <= 1.0
    y = rng.normal(0, 1, 1_000_000) * 10.0 ** rng.integers(-6, 6, 1_000_000)
    xx = rng.normal(0, 1, 1_000_000) * 10.0 ** rng.integers(-6, 6, 1_000_000)
    assert _ulps(O.dm_eval(O.DM_ATAN2, xx, y), np.arctan2(y, xx)).max() <= 1.0
    x = np.concatenate([rng.uniform(-1, 1, 500_000), 1.0 - 10.0 ** rng.uniform(-16, 0, 100_000)])
    assert _ulps(O.dm_eval(O.DM_ACOS, x), np.arccos(x)).max() <= 2.0
    # huge headings lose accuracy gracefully (folded by 2 pi in plain arithmetic first), never garbage
    x = rng.uniform(-1e9, 1e9, 100_000)
    assert np.abs(O.dm_eval(O.DM_SIN, x) - np.sin(x)).max() < 1e-6


def test_special_values():
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-310, -1e-310, 1e300, -1e300, 5e-324])
    Y, X = np.meshgrid(sp, sp)
    with np.errstate(all="ignore"):
        r, e = O.dm_eval(O.DM_ATAN2, X.ravel(), Y.ravel()), np.arctan2(Y.ravel(), X.ravel())
    ok = (np.isnan(r) & np.isnan(e)) | ((_ulps(r, e) <= 1.0) & (np.signbit(r) == np.signbit(e)))
    assert ok.all(), (Y.ravel()[~ok], X.ravel()[~ok], r[~ok], e[~ok])
    ac = O.dm_eval(O.DM_ACOS, np.array([1.0, -1.0, 0.0, 1.0000000000000002, -1.5, np.nan]))
    assert ac[0] == 0.0 and ac[1] == math.pi and ac[2] == math.pi / 2 and np.isnan(ac[3:]).all()
    s = O.dm_eval(O.DM_SIN, np.array([0.0, math.pi / 2, math.pi, np.inf, np.nan]))
    c = O.dm_eval(O.DM_COS, np.array([0.0, math.pi / 2, math.pi, np.inf, np.nan]))
    assert s[0] == 0.0 and s[1] == 1.0 and c[0] == 1.0 and c[2] == -1.0 and np.isnan(s[3:]).all() and np.isnan(c[3:]).all()
    assert s[2] == 1.2246467991473532e-16 and c[1] == 6.123233995736766e-17       # sin / cos of the doubles nearest pi, pi/2
    # the branch the Dubins words turn on: identical operands give a turn of exactly 0, never a negative epsilon
    a = O.dm_eval(O.DM_ATAN2, np.array([3.0, -2.5]), np.array([1.25, 0.75]))
    assert (a - a == 0.0).all()


def test_libm_build_agrees_within_rounding_in_general_position():
    L = O.libm_variant()
    x = np.array([0.3]); out = np.empty(1)
    assert L.orc_dm_eval(0, O._dp(x), O._dp(x), 1, O._dp(out)) == 1          # really the cross-check build
    rng = np.random.default_rng(9)
    n = 3000
    s = np.c_[rng.uniform(-50, 50, (n, 2)), np.zeros(n), rng.uniform(0, 2 * math.pi, n)]
    g = s.copy(); g[:, :2] += rng.normal(0, 6.0, (n, 2)); g[:, 3] = rng.uniform(0, 2 * math.pi, n)
    len_diff = 0
    for i in range(n):
        for r_min in (1.0, 2.0):
            c, w, traj = O.dubins_steer(s[i], g[i], r_min)
            cost = C.c_double(); word = C.create_string_buffer(4); tj = np.zeros((1024, 2)); tl = C.c_int()
            L.orc_dubins_steer(O._dp(s[i]), O._dp(g[i]), r_min, C.byref(cost), word, O._dp(tj), 1024, C.byref(tl))
            assert abs(cost.value - c) <= 1e-12 * max(1.0, abs(c)) and word.value.decode() == w, (i, c, cost.value)
            if tl.value != len(traj):            # an arc whose end rounds across a multiple of 0.1 rad
                len_diff += 1
                continue
            assert np.abs(tj[: tl.value] - traj).max() <= 1e-12 * 60.0, i
    assert len_diff <= 2
