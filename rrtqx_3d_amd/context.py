"""numpy-facing wrapper of one rrtx_ctx (one tree + obstacle lists on one GPU).

Thin by design: every method is one C-ABI call of include/rrtx.h.  The
reference-named API (kdInsert, kdFindWithinRange, explicitEdgeCheck, ...) lives in
rrtqx_3d_amd/drrt.py on top of this.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi
from ._capi import RrtxError, Stats, f64


class Context:
    def __init__(self, dim: int = 3, device: int = 0, node_capacity: int = 1024):
        self._lib = _capi.load()
        _capi.verify_runtime()      # one HIP runtime image per process, or refuse
        self.dim = dim
        self.device = device
        h = C.c_void_p()
        rc = self._lib.rrtx_create(C.byref(h), dim, device, node_capacity)
        if rc != _capi.RRTX_OK:
            raise RrtxError(rc, self._lib.rrtx_create_error().decode())
        self._h = h

    # ---- lifetime -------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.rrtx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc: int):
        if rc != _capi.RRTX_OK:
            raise RrtxError(rc, self._lib.rrtx_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_stream(self, stream_ptr: Optional[int]):
        self._check(self._lib.rrtx_set_stream(self._h, stream_ptr))

    def get_stream(self) -> int:
        return self._lib.rrtx_get_stream(self._h) or 0

    def sync(self):
        self._check(self._lib.rrtx_sync(self._h))

    def profile(self, level):
        """0/False off, 1 range-scan kernel only, 2/True every kernel family."""
        lvl = 2 if level is True else int(level)
        self._check(self._lib.rrtx_profile(self._h, lvl))

    def set_option(self, option: int, value: int):
        self._check(self._lib.rrtx_set_option(self._h, option, value))

    def get_option(self, option: int) -> int:
        v = C.c_int64()
        self._check(self._lib.rrtx_get_option(self._h, option, C.byref(v)))
        return int(v.value)

    @property
    def space_has_time(self) -> bool:
        """CSpace.spaceHasTime as the CONTEXT holds it (no shadow copy: buffers are sized from this)."""
        return self.get_option(_capi.RRTX_OPT_SPACE_HAS_TIME) != 0

    def stats(self) -> Stats:
        s = Stats()
        self._check(self._lib.rrtx_stats(self._h, C.byref(s)))
        return s

    # ---- tree -------------------------------------------------------------------
    def nodes_append(self, pos) -> int:
        pos = f64(pos, (-1, self.dim))
        first = C.c_int64()
        self._check(self._lib.rrtx_nodes_append(self._h, _capi._ptr(pos), pos.shape[0], C.byref(first)))
        return first.value

    def nodes_append_dev(self, dev_ptr: int, n: int):
        self._check(self._lib.rrtx_nodes_append_dev(self._h, dev_ptr, n))

    @property
    def n_nodes(self) -> int:
        return self._lib.rrtx_nodes_count(self._h)

    def set_wrap(self, dim_index: int, period: float):
        self._check(self._lib.rrtx_set_wrap(self._h, dim_index, period))

    # ---- obstacles ----------------------------------------------------------------
    def spheres_set(self, cxyzr, active=None):
        cxyzr = f64(cxyzr, (-1, 4))
        act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        self._check(self._lib.rrtx_spheres_set(self._h, _capi._ptr(cxyzr), _capi._ptr(act), cxyzr.shape[0]))

    def obstacle_update(self, which: int, radius: float, active: bool):
        self._check(self._lib.rrtx_obstacle_update(self._h, which, radius, 1 if active else 0))

    def polygons_set(self, polys, kinds=None, active=None, centre_radius=None, paths=None):
        """polys: list of (P_i x 2) vertex arrays in list order; paths: per obstacle an (M_i x 3) array of
        (dx, dy, t) rows for the moving kinds 6 / 7 (None or empty for the others)."""
        polys = [f64(p, (-1, 2)) for p in polys]
        m = len(polys)
        off = np.zeros(m + 1, dtype=np.int32)
        for i, p in enumerate(polys):
            off[i + 1] = off[i] + p.shape[0]
        vxy = np.concatenate(polys, axis=0) if m else np.zeros((0, 2))
        vxy = f64(vxy, (-1, 2))
        k = None if kinds is None else np.ascontiguousarray(kinds, dtype=np.uint8)
        a = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
        cr = None if centre_radius is None else f64(centre_radius, (-1, 3))
        self._check(self._lib.rrtx_polygons_set(self._h, _capi._ptr(off), _capi._ptr(vxy), _capi._ptr(cr),
                                                _capi._ptr(k), _capi._ptr(a), m))
        if paths is not None:
            self.polygon_paths_set(paths)

    def polygon_paths_set(self, paths):
        """Obstacle.path of every polygon obstacle (list order); see rrtx_polygon_paths_set."""
        rows = [np.zeros((0, 3)) if p is None else f64(p, (-1, 3)) for p in paths]
        m = len(rows)
        off = np.zeros(m + 1, dtype=np.int32)
        for i, p in enumerate(rows):
            off[i + 1] = off[i] + p.shape[0]
        xyt = f64(np.concatenate(rows, axis=0) if m else np.zeros((0, 3)), (-1, 3))
        self._check(self._lib.rrtx_polygon_paths_set(self._h, _capi._ptr(off), _capi._ptr(xyt), m))

    # ---- nearest neighbours ---------------------------------------------------------
    def nn_nearest(self, q) -> Tuple[np.ndarray, np.ndarray]:
        q = f64(q, (-1, self.dim))
        nq = q.shape[0]
        idx = np.empty(nq, dtype=np.int32)
        dist = np.empty(nq, dtype=np.float64)
        self._check(self._lib.rrtx_nn_nearest(self._h, _capi._ptr(q), nq, _capi._ptr(idx), _capi._ptr(dist)))
        return idx, dist

    def nn_knearest(self, q, k: int):
        """kdFindKNearest per query: (idx, dist, count); rows are max(k, 2) wide, sorted by
        (distance, index), count[i] entries of row i are filled."""
        q = f64(q, (-1, self.dim))
        nq = q.shape[0]
        stride = max(int(k), 2)
        idx = np.empty((nq, stride), dtype=np.int32)
        dist = np.empty((nq, stride), dtype=np.float64)
        count = np.empty(nq, dtype=np.int32)
        self._check(self._lib.rrtx_nn_knearest(self._h, _capi._ptr(q), nq, int(k), _capi._ptr(idx), _capi._ptr(dist),
                                               _capi._ptr(count)))
        return idx, dist, count

    def nn_radius(self, q, r, cap: Optional[int] = None):
        """Returns CSR (offsets[nq+1], idx, dist). r: scalar or per-query array."""
        q = f64(q, (-1, self.dim))
        nq = q.shape[0]
        r_arr = f64(r, (-1,))
        stride = 0 if r_arr.shape[0] == 1 else 1
        if stride == 1 and r_arr.shape[0] != nq:
            raise ValueError("r must be a scalar or have one entry per query")
        if cap is None:
            cap = max(64 * nq, 1024)
        offsets = np.empty(nq + 1, dtype=np.int64)
        while True:
            idx = np.empty(cap, dtype=np.int32)
            dist = np.empty(cap, dtype=np.float64)
            needed = C.c_int64()
            rc = self._lib.rrtx_nn_radius(self._h, _capi._ptr(q), _capi._ptr(r_arr), stride, nq,
                                          _capi._ptr(offsets), _capi._ptr(idx), _capi._ptr(dist), cap,
                                          C.byref(needed))
            if rc == _capi.RRTX_E_CAPACITY:
                cap = int(needed.value)     # two-call pattern
                continue
            self._check(rc)
            n = int(needed.value)
            return offsets, idx[:n], dist[:n]

    # ---- collision ---------------------------------------------------------------------
    def edges_check(self, p0, p1, robot_radius: float, kind: int = 0, obstacle: int = -1,
                    want_first: bool = True):
        p0 = f64(p0, (-1, self.dim))
        p1 = f64(p1, (-1, self.dim))
        ne = p0.shape[0]
        hit = np.empty(ne, dtype=np.uint8)
        first = np.empty(ne, dtype=np.int32) if want_first else None
        self._check(self._lib.rrtx_edges_check(self._h, kind, _capi._ptr(p0), _capi._ptr(p1), ne, robot_radius,
                                               obstacle, _capi._ptr(hit), _capi._ptr(first)))
        return hit, first

    def points_check(self, p, robot_radius: float, kind: int = 0, quick: bool = True, want_clearance: bool = True):
        p = f64(p, (-1, self.dim))
        n = p.shape[0]
        unsafe = np.empty(n, dtype=np.uint8)
        clr = np.empty(n, dtype=np.float64) if want_clearance else None
        self._check(self._lib.rrtx_points_check(self._h, kind, _capi._ptr(p), n, robot_radius, 1 if quick else 0,
                                                _capi._ptr(unsafe), _capi._ptr(clr) if want_clearance else None))
        return unsafe, clr

    # ---- steering -------------------------------------------------------------------------
    def simple_steer(self, s, g):
        s = f64(s, (-1, self.dim))
        g = f64(g, (-1, self.dim))
        ne = s.shape[0]
        dist = np.empty(ne, dtype=np.float64)
        wdist = np.empty(ne, dtype=np.float64)
        self._check(self._lib.rrtx_simple_steer(self._h, _capi._ptr(s), _capi._ptr(g), ne, _capi._ptr(dist),
                                                _capi._ptr(wdist)))
        return dist, wdist

    def dubins_steer(self, s, g, r_min: float):
        s = f64(s, (-1, 4))
        g = f64(g, (-1, 4))
        ne = s.shape[0]
        cost = np.empty(ne, dtype=np.float64)
        word = np.empty((ne, 3), dtype=np.uint8)
        self._check(self._lib.rrtx_dubins_steer(self._h, _capi._ptr(s), _capi._ptr(g), ne, r_min, _capi._ptr(cost),
                                                _capi._ptr(word)))
        return cost, word.view("S3").ravel()      # numpy bytes array: b"rsl", b"rsr", ...

    def set_space_has_time(self, has_time: bool):
        """CSpace.spaceHasTime for the Dubins entry points (dim = 4: [x y t theta])."""
        self.set_option(_capi.RRTX_OPT_SPACE_HAS_TIME, 1 if has_time else 0)

    def set_dubins_velocity(self, v_min: float, v_max: float):
        """S.dubinsMinVelocity / S.dubinsMaxVelocity (validMove in a space with time)."""
        self._check(self._lib.rrtx_set_dubins_velocity(self._h, float(v_min), float(v_max)))

    def dubins_steer_full(self, s, g, r_min: float):
        """calculateTrajectory's scalars: dict(dist, wdist, velocity, word, valid_move)."""
        s = f64(s, (-1, 4))
        g = f64(g, (-1, 4))
        ne = s.shape[0]
        dist = np.empty(ne, dtype=np.float64)
        wdist = np.empty(ne, dtype=np.float64)
        vel = np.empty(ne, dtype=np.float64)
        word = np.empty((ne, 3), dtype=np.uint8)
        valid = np.empty(ne, dtype=np.uint8)
        self._check(self._lib.rrtx_dubins_steer_full(self._h, _capi._ptr(s), _capi._ptr(g), ne, r_min, _capi._ptr(dist),
                                                     _capi._ptr(wdist), _capi._ptr(vel), _capi._ptr(word),
                                                     _capi._ptr(valid)))
        return dict(dist=dist, wdist=wdist, velocity=vel, word=word.view("S3").ravel(), valid_move=valid)

    def dubins_edges_check(self, s, g, r_min: float, robot_radius: float):
        s = f64(s, (-1, 4))
        g = f64(g, (-1, 4))
        ne = s.shape[0]
        cost = np.empty(ne, dtype=np.float64)
        word = np.empty((ne, 3), dtype=np.uint8)
        hit = np.empty(ne, dtype=np.uint8)
        tl = np.empty(ne, dtype=np.int32)
        self._check(self._lib.rrtx_dubins_edges_check(self._h, _capi._ptr(s), _capi._ptr(g), ne, r_min,
                                                      robot_radius, _capi._ptr(cost), _capi._ptr(word),
                                                      _capi._ptr(hit), _capi._ptr(tl)))
        return cost, word.view("S3").ravel(), hit, tl

    def dubins_trajectory(self, s, g, r_min: float):
        """edge.trajectory of every Dubins edge: (traj_off[ne+1] in rows, traj[rows, 2]) -- rows of
        (x, y, t) in a space with time (set_space_has_time)."""
        s = f64(s, (-1, 4))
        g = f64(g, (-1, 4))
        ne = s.shape[0]
        off = np.empty(ne + 1, dtype=np.int64)
        cap = max(64 * ne, 64)
        cols = 3 if self.space_has_time else 2
        while True:
            xy = np.empty((cap, cols), dtype=np.float64)
            needed = C.c_int64()
            rc = self._lib.rrtx_dubins_trajectory(self._h, _capi._ptr(s), _capi._ptr(g), ne, r_min, _capi._ptr(off),
                                                  _capi._ptr(xy), cols, cap, C.byref(needed))
            if rc == _capi.RRTX_E_CAPACITY:
                cap = int(needed.value)
                continue
            self._check(rc)
            return off, xy[: int(needed.value)]

    def detmath_eval(self, op: int, x, y=None):
        """include/rrtx_detmath.h on the device, element-wise: 0 sin(x), 1 cos(x), 2 atan2(y, x), 3 acos(x)."""
        x = f64(x, (-1,))
        y = x if y is None else f64(y, (-1,))
        out = np.empty_like(x)
        self._check(self._lib.rrtx_detmath_eval(self._h, op, _capi._ptr(x), _capi._ptr(y), x.size, _capi._ptr(out)))
        return out

    # ---- fused extend() preamble --------------------------------------------------------------
    def host_register(self, arr: np.ndarray):
        """Page-lock a caller array that outlives many calls (rrtx_host_register): output pointers inside it receive
        their results by direct DMA instead of through the context's staging arena."""
        assert arr.flags["C_CONTIGUOUS"]
        self._check(self._lib.rrtx_host_register(self._h, arr.ctypes.data, arr.nbytes))

    def host_unregister(self, arr: np.ndarray):
        self._check(self._lib.rrtx_host_unregister(self._h, arr.ctypes.data))

    def extend_out_buffers(self, nq: int, cap: int, register: bool = False) -> dict:
        """Output arrays for extend_candidates(..., out=...) that a caller keeps across calls, as a Julia host does
        (one allocation; register=True page-locks them)."""
        out = dict(offsets=np.empty(nq + 1, dtype=np.int64), idx=np.empty(cap, dtype=np.int32),
                   cost=np.empty(cap, dtype=np.float64), hit_out=np.empty(cap, dtype=np.uint8),
                   hit_in=np.empty(cap, dtype=np.uint8), nearest_idx=np.empty(nq, dtype=np.int32),
                   nearest_dist=np.empty(nq, dtype=np.float64), sample_unsafe=np.empty(nq, dtype=np.uint8))
        if register:
            for a in out.values():
                self.host_register(a)
        return out

    def extend_candidates(self, q, r: float, robot_radius: float, cap: Optional[int] = None, out: Optional[dict] = None):
        q = f64(q, (-1, self.dim))
        nq = q.shape[0]
        if out is not None:                      # caller-owned arrays (extend_out_buffers): no allocation, no growth
            cap = out["idx"].shape[0]
            needed = C.c_int64()
            self._check(self._lib.rrtx_extend_candidates(
                self._h, _capi._ptr(q), nq, r, robot_radius, _capi._ptr(out["offsets"]), _capi._ptr(out["idx"]),
                _capi._ptr(out["cost"]), _capi._ptr(out["hit_out"]), _capi._ptr(out["hit_in"]), cap, C.byref(needed),
                _capi._ptr(out["nearest_idx"]), _capi._ptr(out["nearest_dist"]), _capi._ptr(out["sample_unsafe"])))
            n = int(needed.value)
            return dict(offsets=out["offsets"], idx=out["idx"][:n], cost=out["cost"][:n], hit_out=out["hit_out"][:n],
                        hit_in=out["hit_in"][:n], nearest_idx=out["nearest_idx"], nearest_dist=out["nearest_dist"],
                        sample_unsafe=out["sample_unsafe"])
        if cap is None:
            cap = max(64 * nq, 1024)
        offsets = np.empty(nq + 1, dtype=np.int64)
        nidx = np.empty(nq, dtype=np.int32)
        ndist = np.empty(nq, dtype=np.float64)
        unsafe = np.empty(nq, dtype=np.uint8)
        while True:
            idx = np.empty(cap, dtype=np.int32)
            cost = np.empty(cap, dtype=np.float64)
            hout = np.empty(cap, dtype=np.uint8)
            hin = np.empty(cap, dtype=np.uint8)
            needed = C.c_int64()
            rc = self._lib.rrtx_extend_candidates(self._h, _capi._ptr(q), nq, r, robot_radius, _capi._ptr(offsets),
                                                  _capi._ptr(idx), _capi._ptr(cost), _capi._ptr(hout),
                                                  _capi._ptr(hin), cap, C.byref(needed), _capi._ptr(nidx),
                                                  _capi._ptr(ndist), _capi._ptr(unsafe))
            if rc == _capi.RRTX_E_CAPACITY:
                cap = int(needed.value)
                continue
            self._check(rc)
            n = int(needed.value)
            return dict(offsets=offsets, idx=idx[:n], cost=cost[:n], hit_out=hout[:n], hit_in=hin[:n],
                        nearest_idx=nidx, nearest_dist=ndist, sample_unsafe=unsafe)

    def extend_candidates_dubins(self, q, r: float, robot_radius: float, r_min: float, cap: Optional[int] = None):
        """Fused extend() preamble for Edge = DubinsEdge (dim 4, theta wrapped, polygon obstacles)."""
        q = f64(q, (-1, 4))
        nq = q.shape[0]
        if cap is None:
            cap = max(256 * nq, 1024)
        offsets = np.empty(nq + 1, dtype=np.int64)
        nidx = np.empty(nq, dtype=np.int32)
        ndist = np.empty(nq, dtype=np.float64)
        unsafe = np.empty(nq, dtype=np.uint8)
        while True:
            idx = np.empty(cap, dtype=np.int32)
            key = np.empty(cap, dtype=np.float64)
            co = np.empty(cap, dtype=np.float64)
            ci = np.empty(cap, dtype=np.float64)
            wo = np.empty((cap, 3), dtype=np.uint8)
            wi = np.empty((cap, 3), dtype=np.uint8)
            ho = np.empty(cap, dtype=np.uint8)
            hi = np.empty(cap, dtype=np.uint8)
            needed = C.c_int64()
            rc = self._lib.rrtx_extend_candidates_dubins(
                self._h, _capi._ptr(q), nq, r, robot_radius, r_min, _capi._ptr(offsets), _capi._ptr(idx),
                _capi._ptr(key), _capi._ptr(co), _capi._ptr(ci), _capi._ptr(wo), _capi._ptr(wi), _capi._ptr(ho),
                _capi._ptr(hi), cap, C.byref(needed), _capi._ptr(nidx), _capi._ptr(ndist), _capi._ptr(unsafe))
            if rc == _capi.RRTX_E_CAPACITY:
                cap = int(needed.value)
                continue
            self._check(rc)
            n = int(needed.value)
            return dict(offsets=offsets, idx=idx[:n], key=key[:n], cost_out=co[:n], cost_in=ci[:n],
                        word_out=wo[:n].view("S3").ravel(), word_in=wi[:n].view("S3").ravel(), hit_out=ho[:n],
                        hit_in=hi[:n], nearest_idx=nidx, nearest_dist=ndist, sample_unsafe=unsafe)

    # ---- device-pointer variants (pointers are ints, e.g. torch.Tensor.data_ptr()) ----------------
    def nn_nearest_dev(self, q_ptr: int, nq: int, idx_ptr: int, dist_ptr: int):
        self._check(self._lib.rrtx_nn_nearest_dev(self._h, q_ptr, nq, idx_ptr, dist_ptr))

    def nn_knearest_dev(self, q_ptr: int, nq: int, k: int, idx_ptr: int, dist_ptr: int, count_ptr: int):
        self._check(self._lib.rrtx_nn_knearest_dev(self._h, q_ptr, nq, k, idx_ptr, dist_ptr, count_ptr))

    def nn_radius_dev(self, q_ptr: int, r: float, nq: int, offsets_ptr: int, idx_ptr: int, dist_ptr: int, cap: int,
                      needed_ptr: int):
        self._check(self._lib.rrtx_nn_radius_dev(self._h, q_ptr, r, nq, offsets_ptr, idx_ptr, dist_ptr, cap,
                                                 needed_ptr))

    def edges_check_dev(self, kind: int, p0_ptr: int, p1_ptr: int, ne: int, robot_radius: float, obstacle: int,
                        obs_begin: int, obs_end: int, hit_ptr: int, first_ptr: Optional[int]):
        self._check(self._lib.rrtx_edges_check_dev(self._h, kind, p0_ptr, p1_ptr, ne, robot_radius, obstacle,
                                                   obs_begin, obs_end, hit_ptr, first_ptr))

    def points_check_dev(self, kind: int, p_ptr: int, n: int, robot_radius: float, quick: bool, unsafe_ptr: int,
                         clr_ptr: Optional[int]):
        self._check(self._lib.rrtx_points_check_dev(self._h, kind, p_ptr, n, robot_radius, 1 if quick else 0,
                                                    unsafe_ptr, clr_ptr))

    def extend_candidates_dev(self, q_ptr: int, nq: int, r: float, robot_radius: float, offsets_ptr: int,
                              idx_ptr: int, cost_ptr: int, hit_out_ptr: int, hit_in_ptr: int, cap: int,
                              needed_ptr: int, nearest_idx_ptr: Optional[int] = None,
                              nearest_dist_ptr: Optional[int] = None, unsafe_ptr: Optional[int] = None):
        self._check(self._lib.rrtx_extend_candidates_dev(self._h, q_ptr, nq, r, robot_radius, offsets_ptr, idx_ptr,
                                                         cost_ptr, hit_out_ptr, hit_in_ptr, cap, needed_ptr,
                                                         nearest_idx_ptr, nearest_dist_ptr, unsafe_ptr))

    def extend_candidates_dubins_dev(self, q_ptr: int, nq: int, r: float, robot_radius: float, r_min: float,
                                     offsets_ptr: int, idx_ptr: int, key_ptr: int, cost_out_ptr: int, cost_in_ptr: int,
                                     word_out_ptr: Optional[int], word_in_ptr: Optional[int], hit_out_ptr: int,
                                     hit_in_ptr: int, cap: int, needed_ptr: int, nearest_idx_ptr: Optional[int] = None,
                                     nearest_dist_ptr: Optional[int] = None, unsafe_ptr: Optional[int] = None):
        self._check(self._lib.rrtx_extend_candidates_dubins_dev(
            self._h, q_ptr, nq, r, robot_radius, r_min, offsets_ptr, idx_ptr, key_ptr, cost_out_ptr, cost_in_ptr,
            word_out_ptr, word_in_ptr, hit_out_ptr, hit_in_ptr, cap, needed_ptr, nearest_idx_ptr, nearest_dist_ptr,
            unsafe_ptr))

    def pack_hits_dev(self, hit_out_ptr: int, hit_in_ptr: int, n_valid_ptr: int, cap: int, words_ptr: int):
        self._check(self._lib.rrtx_pack_hits_dev(self._h, hit_out_ptr, hit_in_ptr, n_valid_ptr, cap, words_ptr))

    # ---- obstacle sweeps over the device mirror of the planner's edges ----------------
    def graph_edges_append(self, start_idx, end_idx) -> int:
        a = np.ascontiguousarray(start_idx, dtype=np.int32).reshape(-1)
        b = np.ascontiguousarray(end_idx, dtype=np.int32).reshape(-1)
        assert a.shape == b.shape
        first = C.c_int64()
        self._check(self._lib.rrtx_graph_edges_append(self._h, _capi._ptr(a), _capi._ptr(b), a.shape[0], C.byref(first)))
        return first.value

    @property
    def n_graph_edges(self) -> int:
        return int(self._lib.rrtx_graph_edges_count(self._h))

    def graph_edges_clear(self):
        self._check(self._lib.rrtx_graph_edges_clear(self._h))

    def graph_edges_set_dist(self, first_id: int, dist):
        """edge.dist of mirrored edges [first_id, first_id + len(dist)) (Inf = blocked)"""
        d = np.ascontiguousarray(dist, dtype=np.float64).reshape(-1)
        self._check(self._lib.rrtx_graph_edges_set_dist(self._h, int(first_id), _capi._ptr(d), d.shape[0]))

    def graph_edges_block(self, edge_ids):
        """addNewObstacle's `edge.dist = Inf` for the ids an obstacle sweep returned"""
        ids = np.ascontiguousarray(edge_ids, dtype=np.int32).reshape(-1)
        self._check(self._lib.rrtx_graph_edges_block(self._h, _capi._ptr(ids), ids.shape[0]))

    def graph_cost_to_root(self, root_idx: int, want_parent: bool = True, update: bool = False):
        """rrtLMC of every node at the fixed point of rewire / reduceInconsistency (changeThresh = 0) over the
        edge mirror, and the id of each node's parent edge (-1: root or orphan).  Returns (lmc, parent_edge, passes).
        update=True continues from the previous solve (rrtx_graph_cost_update): same answer, less work."""
        n = self.n_nodes
        lmc = np.empty(n, dtype=np.float64)
        par = np.empty(n, dtype=np.int32) if want_parent else None
        passes = C.c_int32()
        fn = self._lib.rrtx_graph_cost_update if update else self._lib.rrtx_graph_cost_to_root
        self._check(fn(self._h, int(root_idx), _capi._ptr(lmc), _capi._ptr(par) if want_parent else None, C.byref(passes)))
        return lmc, par, passes.value

    def graph_cost_update(self, root_idx: int, want_parent: bool = True):
        return self.graph_cost_to_root(root_idx, want_parent, update=True)

    def obstacle_sweep(self, obstacle: int, search_range: float, robot_radius: float, cap: Optional[int] = None):
        """addNewObstacle's edge loop: ids (ascending) of the registered edges that start within
        search_range of sphere `obstacle` and collide with it."""
        if cap is None:
            cap = 4096
        while True:
            ids = np.empty(max(cap, 1), dtype=np.int32)
            needed = C.c_int64()
            rc = self._lib.rrtx_obstacle_sweep(self._h, obstacle, search_range, robot_radius, _capi._ptr(ids), cap,
                                               C.byref(needed))
            if rc == _capi.RRTX_E_CAPACITY:
                cap = int(needed.value)
                continue
            self._check(rc)
            return ids[:int(needed.value)]

    def obstacle_sweep_polygon(self, obstacle: int, robot_radius: float, delta: float, r_min: float = 0.0,
                               remove: bool = False, cap: Optional[int] = None):
        """The edge loops of addNewObstacle / removeObstacle for the polygon list (R/DRRT.jl:3048-3290): ids
        (ascending) of the mirrored edges that start at a node in conflict with polygon `obstacle` and collide with it
        (remove: that are blocked, collide with it and with no other obstacle in use).  SimpleEdge in a dim = 3
        context, DubinsEdge (r_min) in a dim = 4 one."""
        if cap is None:
            cap = 4096
        while True:
            ids = np.empty(max(cap, 1), dtype=np.int32)
            needed = C.c_int64()
            rc = self._lib.rrtx_obstacle_sweep_polygon(self._h, obstacle, robot_radius, delta, r_min, 1 if remove else 0,
                                                       _capi._ptr(ids), cap, C.byref(needed))
            if rc == _capi.RRTX_E_CAPACITY:
                cap = int(needed.value)
                continue
            self._check(rc)
            return ids[:int(needed.value)]

    def dubins_edges_check_obstacle(self, s, g, r_min: float, robot_radius: float, obstacle: int):
        """explicitEdgeCheck(S, edge::DubinsEdge, ob) against polygon `obstacle` alone."""
        s = f64(s, (-1, 4))
        g = f64(g, (-1, 4))
        hit = np.empty(s.shape[0], dtype=np.uint8)
        self._check(self._lib.rrtx_dubins_edges_check_obstacle(self._h, _capi._ptr(s), _capi._ptr(g), s.shape[0], r_min,
                                                               robot_radius, obstacle, _capi._ptr(hit)))
        return hit

    def edges_check_idx(self, start_idx, end_idx, robot_radius: float, obstacle: int = -1, obstacle_mask=None,
                        want_first: bool = True):
        """Edges as node-index pairs (obstacle sweeps, R/DRRT_Q.jl:3220-3362)."""
        s = np.ascontiguousarray(start_idx, dtype=np.int32)
        e = np.ascontiguousarray(end_idx, dtype=np.int32)
        ne = s.shape[0]
        hit = np.empty(ne, dtype=np.uint8)
        first = np.empty(ne, dtype=np.int32) if want_first else None
        mask = None if obstacle_mask is None else np.ascontiguousarray(obstacle_mask, dtype=np.uint8)
        self._check(self._lib.rrtx_edges_check_idx(self._h, _capi._ptr(s), _capi._ptr(e), ne, robot_radius, obstacle,
                                                   _capi._ptr(mask), _capi._ptr(hit), _capi._ptr(first)))
        return hit, first
