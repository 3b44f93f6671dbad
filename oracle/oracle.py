"""ctypes loader for the CPU oracle (oracle/rrtx_oracle.c).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from rrtqx_3d_amd/ (the product path).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "librrtx_oracle.so")
# the same source with glibc's sin / cos / atan2 / acos on the Dubins paths (-DORC_LIBM_TRIG): CPU cross-check only
_LIBM_PATH = os.path.join(_HERE, "_build", "librrtx_oracle_libm.so")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)


class Sphere(C.Structure):
    _fields_ = [("c", C.c_double * 3), ("radius", C.c_double), ("life_span", C.c_double),
                ("unused", C.c_int32), ("pad", C.c_int32)]


class Polygon(C.Structure):
    _fields_ = [("kind", C.c_int32), ("nverts", C.c_int32), ("verts", c_double_p),
                ("cx", C.c_double), ("cy", C.c_double), ("radius", C.c_double),
                ("life_span", C.c_double), ("unused", C.c_int32), ("npath", C.c_int32),
                ("path", c_double_p)]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds). Returns the .so path."""
    src = os.path.join(_HERE, "rrtx_oracle.c")
    src2 = os.path.join(_HERE, "rrtx_oracle_graph.c")
    hdr = os.path.join(_HERE, "rrtx_oracle.h")
    dm = os.path.join(_HERE, "..", "include", "rrtx_detmath.h")
    stale = (not os.path.exists(_LIB_PATH) or not os.path.exists(_LIBM_PATH)
             or any(os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in (src, src2, hdr, dm)))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.orc_euclid.restype = C.c_double
    L.orc_euclid.argtypes = [c_double_p, c_double_p, C.c_int]
    L.orc_ball_radius.restype = C.c_double
    L.orc_ball_radius.argtypes = [C.c_double, C.c_double, C.c_int64, C.c_int]
    L.orc_kd_create.restype = C.c_void_p
    L.orc_kd_create.argtypes = [C.c_int]
    L.orc_kd_destroy.argtypes = [C.c_void_p]
    L.orc_kd_set_wraps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), c_double_p]
    L.orc_kd_insert.restype = C.c_int64
    L.orc_kd_insert.argtypes = [C.c_void_p, c_double_p]
    L.orc_kd_size.restype = C.c_int64
    L.orc_kd_size.argtypes = [C.c_void_p]
    L.orc_kd_depth.restype = C.c_int64
    L.orc_kd_depth.argtypes = [C.c_void_p]
    L.orc_kd_nearest.argtypes = [C.c_void_p, c_double_p, c_int64_p, c_double_p]
    L.orc_kd_nearest_naive.argtypes = [C.c_void_p, c_double_p, c_int64_p, c_double_p]
    L.orc_kd_find_within_range.restype = C.c_void_p
    L.orc_kd_find_within_range.argtypes = [C.c_void_p, C.c_double, c_double_p]
    L.orc_kd_find_more_within_range.argtypes = [C.c_void_p, C.c_double, c_double_p, C.c_void_p]
    L.orc_list_length.restype = C.c_int64
    L.orc_list_length.argtypes = [C.c_void_p]
    L.orc_list_read.restype = C.c_int64
    L.orc_list_read.argtypes = [C.c_void_p, C.c_int64, c_int32_p, c_double_p]
    L.orc_kd_empty_range_list.argtypes = [C.c_void_p, C.c_void_p]
    for f in (L.orc_kd_knearest, L.orc_kd_knearest_naive):
        f.restype = C.c_int64
        f.argtypes = [C.c_void_p, C.c_int64, c_double_p, C.c_int64, c_int32_p, c_double_p]
    L.orc_range_naive.restype = C.c_int64
    L.orc_range_naive.argtypes = [C.c_void_p, C.c_double, c_double_p, C.c_int64, c_int32_p, c_double_p]
    L.orc_ghost_points.restype = C.c_int
    L.orc_ghost_points.argtypes = [C.c_void_p, c_double_p, C.c_double, C.c_int, c_double_p]
    L.orc_distance_point_to_segment3.restype = C.c_double
    L.orc_distance_point_to_segment3.argtypes = [c_double_p] * 3
    L.orc_edge_check_sphere.restype = C.c_int
    L.orc_edge_check_sphere.argtypes = [C.POINTER(Sphere), c_double_p, c_double_p, C.c_double]
    L.orc_edge_check_spheres.restype = C.c_int
    L.orc_edge_check_spheres.argtypes = [C.POINTER(Sphere), C.c_int, c_double_p, c_double_p, C.c_double, c_int32_p]
    L.orc_point_check_spheres.restype = C.c_int
    L.orc_point_check_spheres.argtypes = [C.POINTER(Sphere), C.c_int, c_double_p, C.c_double, C.c_int, c_double_p]
    L.orc_polygon_ctor.argtypes = [c_double_p, C.c_int, c_double_p, c_double_p, c_double_p]
    L.orc_dist_sqrd_point_to_segment.restype = C.c_double
    L.orc_dist_sqrd_point_to_segment.argtypes = [c_double_p] * 3
    L.orc_segment_dist_sqrd.restype = C.c_double
    L.orc_segment_dist_sqrd.argtypes = [c_double_p] * 4
    L.orc_point_in_polygon.restype = C.c_int
    L.orc_point_in_polygon.argtypes = [c_double_p, c_double_p, C.c_int]
    L.orc_dist_to_polygon_sqrd.restype = C.c_double
    L.orc_dist_to_polygon_sqrd.argtypes = [c_double_p, c_double_p, C.c_int]
    L.orc_edge_check_polygon.restype = C.c_int
    L.orc_edge_check_polygon.argtypes = [C.POINTER(Polygon), c_double_p, c_double_p, C.c_double]
    L.orc_edge_check_polygons.restype = C.c_int
    L.orc_edge_check_polygons.argtypes = [C.POINTER(Polygon), C.c_int, c_double_p, c_double_p, C.c_double, c_int32_p]
    L.orc_point_check_polygons.restype = C.c_int
    L.orc_point_check_polygons.argtypes = [C.POINTER(Polygon), C.c_int, c_double_p, C.c_double, c_double_p]
    L.orc_dubins_steer.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p, C.c_char_p,
                                   c_double_p, C.c_int, C.POINTER(C.c_int)]
    L.orc_dubins_edge_check_polygons.restype = C.c_int
    L.orc_dubins_edge_check_polygons.argtypes = [C.POINTER(Polygon), C.c_int, c_double_p, c_double_p,
                                                 c_double_p, C.c_int, C.c_double, C.c_double, c_int32_p]
    for f in (L.orc_dubins_steer_time, L.orc_dubins_steer_time_pw):
        f.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p, c_double_p, c_double_p,
                      C.c_char_p, c_double_p, C.c_int, C.POINTER(C.c_int)]
    L.orc_find_points_in_conflict_polygon.restype = C.c_void_p
    L.orc_find_points_in_conflict_polygon.argtypes = [C.c_void_p, C.POINTER(Polygon), C.c_double, C.c_double, C.c_int, C.c_int]
    L.orc_dm_eval.restype = C.c_int
    L.orc_dm_eval.argtypes = [C.c_int, c_double_p, c_double_p, C.c_int64, c_double_p]
    L.orc_dubins_valid_move_time.restype = C.c_int
    L.orc_dubins_valid_move_time.argtypes = [c_double_p, c_double_p, C.c_double, C.c_double, C.c_double]
    L.orc_dubins_edge_check_polygons_time.restype = C.c_int
    L.orc_dubins_edge_check_polygons_time.argtypes = [C.POINTER(Polygon), C.c_int, c_double_p, c_double_p,
                                                      c_double_p, C.c_int, C.c_double, C.c_double, c_int32_p]
    L.orc_graph_create.restype = C.c_void_p
    L.orc_graph_create.argtypes = [C.c_int64]
    L.orc_graph_destroy.argtypes = [C.c_void_p]
    L.orc_graph_add_edge.restype = C.c_int64
    L.orc_graph_add_edge.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_int, C.c_int]
    L.orc_graph_set_node.argtypes = [C.c_void_p, C.c_int64, C.c_double, C.c_double]
    L.orc_graph_set_move_goal.argtypes = [C.c_void_p, C.c_int64, C.c_int]
    L.orc_graph_set_edge_dist.argtypes = [C.c_void_p, C.c_int64, C.c_double]
    for f in (L.orc_graph_lmc, L.orc_graph_tree_cost):
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_int64]
    L.orc_graph_parent_edge.restype = C.c_int64
    L.orc_graph_parent_edge.argtypes = [C.c_void_p, C.c_int64]
    L.orc_graph_queue_length.restype = C.c_int64
    L.orc_graph_queue_length.argtypes = [C.c_void_p]
    L.orc_graph_n_edges.restype = C.c_int64
    L.orc_graph_n_edges.argtypes = [C.c_void_p]
    L.orc_graph_verify_in_queue.argtypes = [C.c_void_p, C.c_int64]
    L.orc_graph_verify_in_os.argtypes = [C.c_void_p, C.c_int64]
    L.orc_graph_make_parent_of.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    L.orc_graph_reduce_inconsistency.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_double]
    L.orc_graph_block_edge.argtypes = [C.c_void_p, C.c_int64]
    L.orc_graph_propagate_descendants.argtypes = [C.c_void_p]
    L.orc_julia_range_len.restype = C.c_int64
    L.orc_julia_range_len.argtypes = [C.c_double] * 3
    L.orc_kd_insert_many.restype = None
    L.orc_kd_insert_many.argtypes = [C.c_void_p, c_double_p, C.c_int64]
    L.orc_extend_batch_spheres.restype = C.c_int64
    L.orc_extend_batch_spheres.argtypes = [C.c_void_p, C.POINTER(Sphere), C.c_int, c_double_p, C.c_int64,
                                           C.c_double, C.c_double, c_int64_p, c_int64_p, c_int64_p]
    L.orc_extend_batch_polygons.restype = C.c_int64
    L.orc_extend_batch_polygons.argtypes = [C.c_void_p, C.POINTER(Polygon), C.c_int, c_double_p, C.c_int64,
                                            C.c_double, C.c_double, c_int64_p, c_int64_p, c_int64_p]
    _lib = L
    return L


def _dp(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(c_double_p)


def _vec(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))


def euclid(x, y) -> float:
    x, y = _vec(x), _vec(y)
    return lib().orc_euclid(_dp(x), _dp(y), len(x))


def ball_radius(delta: float, ball_constant: float, n: int, d: int) -> float:
    return lib().orc_ball_radius(delta, ball_constant, n, d)


class KDTree:
    """The reference's incremental kd-tree (R/kdTree_general.jl)."""

    def __init__(self, d: int, wraps=None, wrap_points=None):
        self.d = d
        self._h = lib().orc_kd_create(d)
        if wraps:
            w = (C.c_int * len(wraps))(*wraps)
            wp = (C.c_double * len(wraps))(*wrap_points)
            lib().orc_kd_set_wraps(self._h, len(wraps), w, wp)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_kd_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def insert(self, pos) -> int:
        p = _vec(pos)
        assert len(p) == self.d
        return lib().orc_kd_insert(self._h, _dp(p))

    def insert_many(self, pts: np.ndarray):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        assert pts.ndim == 2 and pts.shape[1] == self.d
        lib().orc_kd_insert_many(self._h, _dp(pts), pts.shape[0])

    @property
    def size(self) -> int:
        return lib().orc_kd_size(self._h)

    def depth(self) -> int:
        return lib().orc_kd_depth(self._h)

    def nearest(self, q, naive: bool = False):
        q = _vec(q)
        idx = C.c_int64()
        dist = C.c_double()
        f = lib().orc_kd_nearest_naive if naive else lib().orc_kd_nearest
        f(self._h, _dp(q), C.byref(idx), C.byref(dist))
        return idx.value, dist.value

    def within_range(self, r: float, q, more=()):
        """kdFindWithinRange (+ kdFindMoreWithinRange for each extra (r, q) in `more`).
        Returns (idx, key) in list order (front first)."""
        q = _vec(q)
        L = lib()
        lst = L.orc_kd_find_within_range(self._h, r, _dp(q))
        for (r2, q2) in more:
            q2 = _vec(q2)
            L.orc_kd_find_more_within_range(self._h, r2, _dp(q2), lst)
        n = L.orc_list_length(lst)
        idx = np.empty(n, dtype=np.int32)
        key = np.empty(n, dtype=np.float64)
        L.orc_list_read(lst, n, idx.ctypes.data_as(c_int32_p), _dp(key))
        L.orc_kd_empty_range_list(self._h, lst)
        return idx, key

    def knearest(self, k: int, q, naive: bool = False):
        """kdFindKNearest / kdFindKNearestNaive: (idx, key) in heap order; raises where the
        reference does (wrapped space)."""
        q = _vec(q)
        f = lib().orc_kd_knearest_naive if naive else lib().orc_kd_knearest
        cap = max(int(k), 2) + 1
        idx = np.empty(cap, dtype=np.int32)
        key = np.empty(cap, dtype=np.float64)
        n = f(self._h, int(k), _dp(q), cap, idx.ctypes.data_as(c_int32_p), _dp(key))
        if n < 0:
            raise RuntimeError("knn search has not been implimented for wrapped space")
        return idx[:n].copy(), key[:n].copy()

    def range_naive(self, r: float, q):
        q = _vec(q)
        L = lib()
        n = L.orc_range_naive(self._h, r, _dp(q), 0, None, None)
        idx = np.empty(n, dtype=np.int32)
        key = np.empty(n, dtype=np.float64)
        L.orc_range_naive(self._h, r, _dp(q), n, idx.ctypes.data_as(c_int32_p), _dp(key))
        return idx, key

    def ghost_points(self, q, best_dist: float):
        q = _vec(q)
        out = np.empty((64, self.d), dtype=np.float64)
        n = lib().orc_ghost_points(self._h, _dp(q), best_dist, 64, _dp(out))
        return out[:n].copy()


def make_spheres(cxyzr: np.ndarray, active=None, life_span=None):
    cxyzr = np.asarray(cxyzr, dtype=np.float64).reshape(-1, 4)
    m = cxyzr.shape[0]
    arr = (Sphere * max(m, 1))()
    for i in range(m):
        arr[i].c[0], arr[i].c[1], arr[i].c[2] = cxyzr[i, 0], cxyzr[i, 1], cxyzr[i, 2]
        arr[i].radius = cxyzr[i, 3]
        arr[i].life_span = float("inf") if life_span is None else float(life_span[i])
        arr[i].unused = 0 if (active is None or active[i]) else 1
    return arr, m


def distance_point_to_segment3(c, p0, p1) -> float:
    c, p0, p1 = _vec(c), _vec(p0), _vec(p1)
    return lib().orc_distance_point_to_segment3(_dp(c), _dp(p0), _dp(p1))


def edge_check_spheres(spheres, m, p0, p1, robot_radius):
    p0, p1 = _vec(p0), _vec(p1)
    fh = C.c_int32()
    hit = lib().orc_edge_check_spheres(spheres, m, _dp(p0), _dp(p1), robot_radius, C.byref(fh))
    return bool(hit), fh.value


def edges_check_spheres(spheres, m, P0: np.ndarray, P1: np.ndarray, robot_radius: float):
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    n, d = P0.shape
    hit = np.zeros(n, dtype=np.uint8)
    first = np.full(n, -1, dtype=np.int32)
    f = lib().orc_edge_check_spheres
    fh = C.c_int32()
    b0, b1, st = P0.ctypes.data, P1.ctypes.data, d * 8
    for i in range(n):
        hit[i] = f(spheres, m, C.cast(b0 + i * st, c_double_p), C.cast(b1 + i * st, c_double_p),
                   robot_radius, C.byref(fh))
        first[i] = fh.value
    return hit, first


def point_check_spheres(spheres, m, p, robot_radius, quick=True):
    p = _vec(p)
    cl = C.c_double()
    r = lib().orc_point_check_spheres(spheres, m, _dp(p), robot_radius, 1 if quick else 0, C.byref(cl))
    return bool(r), cl.value


def points_check_spheres(spheres, m, P: np.ndarray, robot_radius, quick=True):
    P = np.ascontiguousarray(P, dtype=np.float64)
    n, d = P.shape
    unsafe = np.zeros(n, dtype=np.uint8)
    clr = np.zeros(n, dtype=np.float64)
    cl = C.c_double()
    f = lib().orc_point_check_spheres
    for i in range(n):
        unsafe[i] = f(spheres, m, C.cast(P.ctypes.data + i * d * 8, c_double_p), robot_radius,
                      1 if quick else 0, C.byref(cl))
        clr[i] = cl.value
    return unsafe, clr


def polygon_ctor(verts):
    v = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 2))
    cx, cy, r = C.c_double(), C.c_double(), C.c_double()
    lib().orc_polygon_ctor(_dp(v), v.shape[0], C.byref(cx), C.byref(cy), C.byref(r))
    return cx.value, cy.value, r.value


class PolygonSet:
    """A list of polygon obstacles in list order: kind 3 (static), 1 (ball), 6 / 7 (moving along
    paths[i], rows of (dx, dy, t))."""

    def __init__(self, polys, kinds=None, active=None, paths=None):
        self.verts = [np.ascontiguousarray(np.asarray(p, dtype=np.float64).reshape(-1, 2)) for p in polys]
        self.m = len(self.verts)
        self.arr = (Polygon * max(self.m, 1))()
        self.paths = [None] * self.m
        for i in range(self.m):
            if paths is not None and paths[i] is not None and len(paths[i]):
                self.paths[i] = np.ascontiguousarray(np.asarray(paths[i], dtype=np.float64).reshape(-1, 3))
                self.arr[i].npath = self.paths[i].shape[0]
                self.arr[i].path = _dp(self.paths[i])
        for i, v in enumerate(self.verts):
            cx, cy, r = polygon_ctor(v)
            a = self.arr[i]
            a.kind = 3 if kinds is None else int(kinds[i])
            a.nverts = v.shape[0]
            a.verts = _dp(v)
            a.cx, a.cy, a.radius = cx, cy, r
            a.life_span = float("inf")
            a.unused = 0 if (active is None or active[i]) else 1

    def centre_radius(self) -> np.ndarray:
        return np.array([[self.arr[i].cx, self.arr[i].cy, self.arr[i].radius] for i in range(self.m)],
                        dtype=np.float64).reshape(-1, 3)


def dist_sqrd_point_to_segment(pt, a, b) -> float:
    pt, a, b = _vec(pt), _vec(a), _vec(b)
    return lib().orc_dist_sqrd_point_to_segment(_dp(pt), _dp(a), _dp(b))


def segment_dist_sqrd(pa, pb, qa, qb) -> float:
    pa, pb, qa, qb = _vec(pa), _vec(pb), _vec(qa), _vec(qb)
    return lib().orc_segment_dist_sqrd(_dp(pa), _dp(pb), _dp(qa), _dp(qb))


def point_in_polygon(pt, verts) -> bool:
    pt = _vec(pt)
    v = np.ascontiguousarray(np.asarray(verts, dtype=np.float64).reshape(-1, 2))
    return bool(lib().orc_point_in_polygon(_dp(pt), _dp(v), v.shape[0]))


def edge_check_polygons(ps: PolygonSet, p0, p1, robot_radius):
    p0, p1 = _vec(p0), _vec(p1)
    fh = C.c_int32()
    hit = lib().orc_edge_check_polygons(ps.arr, ps.m, _dp(p0), _dp(p1), robot_radius, C.byref(fh))
    return bool(hit), fh.value


def edges_check_polygons(ps: PolygonSet, P0, P1, robot_radius):
    P0 = np.ascontiguousarray(P0, dtype=np.float64)
    P1 = np.ascontiguousarray(P1, dtype=np.float64)
    n, d = P0.shape
    hit = np.zeros(n, dtype=np.uint8)
    first = np.full(n, -1, dtype=np.int32)
    fh = C.c_int32()
    f = lib().orc_edge_check_polygons
    for i in range(n):
        hit[i] = f(ps.arr, ps.m, C.cast(P0.ctypes.data + i * d * 8, c_double_p),
                   C.cast(P1.ctypes.data + i * d * 8, c_double_p), robot_radius, C.byref(fh))
        first[i] = fh.value
    return hit, first


def point_check_polygons(ps: PolygonSet, p, robot_radius):
    p = _vec(p)
    cl = C.c_double()
    r = lib().orc_point_check_polygons(ps.arr, ps.m, _dp(p), robot_radius, C.byref(cl))
    return bool(r), cl.value


def points_in_conflict_polygon(tree: "KDTree", ps: "PolygonSet", j: int, robot_radius: float, delta: float,
                               has_time: bool, has_theta: bool) -> np.ndarray:
    """findPointsInConflictWithObstacle(S, KD, ob::Obstacle, root) (R/DRRT.jl:3048-3125) for obstacle j of the
    list: the node indices of the range list in list order; raises where the reference does."""
    L = lib()
    lst = L.orc_find_points_in_conflict_polygon(tree._h, C.byref(ps.arr[j]), robot_radius, delta, int(has_time), int(has_theta))
    if not lst:
        raise RuntimeError("this type of obstacle not coded for this type of space")
    n = L.orc_list_length(lst)
    idx = np.empty(n, dtype=np.int32)
    key = np.empty(n, dtype=np.float64)
    L.orc_list_read(lst, n, idx.ctypes.data_as(c_int32_p), _dp(key))
    L.orc_kd_empty_range_list(tree._h, lst)
    return idx


def explicit_edge_check_obstacle(ps: "PolygonSet", j: int, a, b, robot_radius: float, dubins: bool, r_min: float = 0.0,
                                 has_time: bool = False, piecewise_time: bool = True) -> bool:
    """explicitEdgeCheck(S, edge, ob) against ONE obstacle of the list: SimpleEdge -> explicitEdgeCheck2D
    (R/DRRT_SimpleEdge_functions.jl:210-212 with the polygon ob), DubinsEdge -> the two-stage check
    (R/DRRT_DubinsEdge_functions.jl:750-774) on the edge's own trajectory."""
    a, b = _vec(a), _vec(b)
    one = C.byref(ps.arr[j])
    if not dubins:
        return bool(lib().orc_edge_check_polygon(one, _dp(a), _dp(b), robot_radius))
    fh = C.c_int32()
    if has_time:
        d, w, v, wd, tr = dubins_steer_time(a, b, r_min, piecewise=piecewise_time)
        if not np.isfinite(w) or len(tr) == 0:
            tr = np.zeros((0, 3))
        return bool(lib().orc_dubins_edge_check_polygons_time(one, 1, _dp(a), _dp(b), _dp(np.ascontiguousarray(tr)), len(tr),
                                                              robot_radius, r_min, C.byref(fh)))
    c, w, traj = dubins_steer(a, b, r_min)
    r = lib().orc_dubins_edge_check_polygons(one, 1, _dp(a), _dp(b), _dp(np.ascontiguousarray(traj)), len(traj), robot_radius,
                                             r_min, C.byref(fh))
    if r < 0:
        raise RuntimeError("Dubins edges against moving obstacles need the time-parameterised trajectory")
    return bool(r)


def add_new_obstacle_edges(tree: "KDTree", pts: np.ndarray, e_start, e_end, ps: "PolygonSet", j: int, robot_radius: float,
                           delta: float, dubins: bool, r_min: float = 0.0, has_time: bool = False) -> np.ndarray:
    """The edge loop of addNewObstacle (R/DRRT.jl:3127-3200) over a mirror of the planner's directed edges: every
    edge that STARTS at a node of findPointsInConflictWithObstacle's list (its out-neighbour edges and its parent
    edge) and for which explicitEdgeCheck(S, edge, ob) is true -- the edges the reference sets to dist = Inf.
    Ascending edge ids."""
    nodes = set(points_in_conflict_polygon(tree, ps, j, robot_radius, delta, has_time, dubins).tolist())
    out = []
    for e in range(len(e_start)):
        if int(e_start[e]) in nodes and explicit_edge_check_obstacle(ps, j, pts[e_start[e]], pts[e_end[e]], robot_radius, dubins,
                                                                     r_min, has_time):
            out.append(e)
    return np.array(out, dtype=np.int32)


def remove_obstacle_edges(tree: "KDTree", pts: np.ndarray, e_start, e_end, e_dist, ps: "PolygonSet", j: int,
                          robot_radius: float, delta: float, dubins: bool, r_min: float = 0.0, has_time: bool = False) -> np.ndarray:
    """The edge loop of removeObstacle (R/DRRT.jl:3202-3290): edges that start at a node in conflict, are blocked
    (dist == Inf), collide with ob, and with no OTHER obstacle that is in use (the caller's `unused` flags say which
    are; the reference also asks startTime <= timeElapsed <= startTime + lifeSpan, which the caller folds into those
    flags) -- the edges the reference resets to distOriginal.  Ascending edge ids."""
    nodes = set(points_in_conflict_polygon(tree, ps, j, robot_radius, delta, has_time, dubins).tolist())
    out = []
    for e in range(len(e_start)):
        if int(e_start[e]) not in nodes or e_dist[e] != np.inf:
            continue
        a, b = pts[e_start[e]], pts[e_end[e]]
        if not explicit_edge_check_obstacle(ps, j, a, b, robot_radius, dubins, r_min, has_time):
            continue
        other = False
        for k in range(ps.m):
            if k != j and not ps.arr[k].unused and explicit_edge_check_obstacle(ps, k, a, b, robot_radius, dubins, r_min, has_time):
                other = True
                break
        if not other:
            out.append(e)
    return np.array(out, dtype=np.int32)


def dubins_steer(s, g, r_min: float, want_traj: bool = True):
    """Returns (cost, word, traj[P,2])."""
    s, g = _vec(s), _vec(g)
    cost = C.c_double()
    word = C.create_string_buffer(4)
    cap = 1024
    traj = np.zeros((cap, 2), dtype=np.float64)
    n = C.c_int()
    lib().orc_dubins_steer(_dp(s), _dp(g), r_min, C.byref(cost), word, _dp(traj) if want_traj else None,
                           cap, C.byref(n))
    return cost.value, word.value.decode(), traj[: n.value].copy()


def dubins_edge_check_polygons(ps: PolygonSet, s, g, traj, robot_radius, r_min):
    s, g = _vec(s), _vec(g)
    traj = np.ascontiguousarray(traj, dtype=np.float64).reshape(-1, 2)
    fh = C.c_int32()
    hit = lib().orc_dubins_edge_check_polygons(ps.arr, ps.m, _dp(s), _dp(g), _dp(traj), traj.shape[0],
                                               robot_radius, r_min, C.byref(fh))
    return bool(hit), fh.value


def dubins_steer_time(s, g, r_min: float, piecewise: bool = False):
    """calculateTrajectory(S, ::DubinsEdge) with S.spaceHasTime: (dist, Wdist, velocity, word, traj[P,3]).
    piecewise: the time column as the HIP kernels form it (orc_dubins_steer_time_pw) instead of the reference's
    running sum; the two differ by rounding only."""
    s, g = _vec(s), _vec(g)
    dist, wdist, vel = C.c_double(), C.c_double(), C.c_double()
    word = C.create_string_buffer(4)
    cap = 1024
    traj = np.zeros((cap, 3), dtype=np.float64)
    n = C.c_int()
    fn = lib().orc_dubins_steer_time_pw if piecewise else lib().orc_dubins_steer_time
    fn(_dp(s), _dp(g), r_min, C.byref(dist), C.byref(wdist), C.byref(vel), word, _dp(traj), cap, C.byref(n))
    return dist.value, wdist.value, vel.value, word.value.decode(), traj[: n.value].copy()


DM_SIN, DM_COS, DM_ATAN2, DM_ACOS = 0, 1, 2, 3


def dm_eval(op: int, x, y=None) -> np.ndarray:
    """include/rrtx_detmath.h element-wise on the host: sin(x), cos(x), atan2(y, x), acos(x)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    y = x if y is None else np.ascontiguousarray(y, dtype=np.float64).ravel()
    assert x.shape == y.shape
    out = np.empty_like(x)
    rc = lib().orc_dm_eval(op, _dp(x), _dp(y), x.size, _dp(out))
    assert rc == 0, "the default oracle build must not be the libm cross-check build"
    return out


def libm_variant() -> C.CDLL:
    """librrtx_oracle_libm.so: the Dubins functions with glibc's transcendentals (orc_dubins_steer only is bound)."""
    build()
    L = C.CDLL(_LIBM_PATH)
    L.orc_dubins_steer.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p, C.c_char_p,
                                   c_double_p, C.c_int, C.POINTER(C.c_int)]
    L.orc_dm_eval.restype = C.c_int
    L.orc_dm_eval.argtypes = [C.c_int, c_double_p, c_double_p, C.c_int64, c_double_p]
    return L


def dubins_valid_move_time(s, g, velocity: float, v_min: float, v_max: float) -> bool:
    s, g = _vec(s), _vec(g)
    return bool(lib().orc_dubins_valid_move_time(_dp(s), _dp(g), velocity, v_min, v_max))


def dubins_edge_check_polygons_time(ps: "PolygonSet", s, g, traj3, robot_radius, r_min):
    s, g = _vec(s), _vec(g)
    traj3 = np.ascontiguousarray(traj3, dtype=np.float64).reshape(-1, 3)
    fh = C.c_int32()
    hit = lib().orc_dubins_edge_check_polygons_time(ps.arr, ps.m, _dp(s), _dp(g), _dp(traj3), traj3.shape[0],
                                                    robot_radius, r_min, C.byref(fh))
    return bool(hit), fh.value


class Graph:
    """The reference's cost propagation on an index graph (rrtx_oracle_graph.c): rrtLMC / rrtTreeCost per node,
    edges with dist, the rrtXQueue heap and the orphan stack; methods carry the reference's names."""

    def __init__(self, n: int):
        self.n = n
        self._h = C.c_void_p(lib().orc_graph_create(n))

    def __del__(self):
        try:
            lib().orc_graph_destroy(self._h)
        except Exception:
            pass

    def add_edge(self, start: int, end: int, dist: float, initial: bool = False, valid_move: bool = True) -> int:
        return lib().orc_graph_add_edge(self._h, start, end, dist, 1 if initial else 0, 1 if valid_move else 0)

    def set_node(self, v: int, lmc: float, tree_cost: float):
        lib().orc_graph_set_node(self._h, v, lmc, tree_cost)

    def set_move_goal(self, v: int, flag: bool = True):
        lib().orc_graph_set_move_goal(self._h, v, 1 if flag else 0)

    def set_edge_dist(self, e: int, dist: float):
        lib().orc_graph_set_edge_dist(self._h, e, dist)

    def lmc(self):
        L = lib()
        return np.array([L.orc_graph_lmc(self._h, v) for v in range(self.n)])

    def tree_cost(self):
        L = lib()
        return np.array([L.orc_graph_tree_cost(self._h, v) for v in range(self.n)])

    def parent_edge(self):
        L = lib()
        return np.array([L.orc_graph_parent_edge(self._h, v) for v in range(self.n)], dtype=np.int64)

    def queue_length(self) -> int:
        return lib().orc_graph_queue_length(self._h)

    def verifyInQueue(self, v: int):
        lib().orc_graph_verify_in_queue(self._h, v)

    def verifyInOSQueue(self, v: int):
        lib().orc_graph_verify_in_os(self._h, v)

    def makeParentOf(self, new_parent: int, node: int, edge: int):
        lib().orc_graph_make_parent_of(self._h, new_parent, node, edge)

    def reduceInconsistency(self, goal: int, root: int, hyberBallRad: float = float("inf"), changeThresh: float = 0.0):
        lib().orc_graph_reduce_inconsistency(self._h, goal, root, hyberBallRad, changeThresh)

    def blockEdge(self, e: int):
        """addNewObstacle's handling of one edge the new obstacle hits (R/DRRT_Q.jl:3248-3268)"""
        lib().orc_graph_block_edge(self._h, e)

    def propogateDescendants(self):
        lib().orc_graph_propagate_descendants(self._h)


def julia_range_len(start, step, stop) -> int:
    return lib().orc_julia_range_len(start, step, stop)


def extend_batch_polygons(tree: KDTree, ps: "PolygonSet", queries: np.ndarray, r: float, robot_radius: float):
    """CPU-baseline loop against a polygon list. Returns (edges_checked, neighbours, hits, nearest_idx)."""
    q = np.ascontiguousarray(queries, dtype=np.float64)
    nq = q.shape[0]
    nearest = np.empty(nq, dtype=np.int64)
    nn, nh = C.c_int64(), C.c_int64()
    e = lib().orc_extend_batch_polygons(tree.handle, ps.arr, ps.m, _dp(q), nq, r, robot_radius,
                                        nearest.ctypes.data_as(c_int64_p), C.byref(nn), C.byref(nh))
    return e, nn.value, nh.value, nearest


def extend_batch_spheres(tree: KDTree, spheres, m, queries: np.ndarray, r: float, robot_radius: float):
    """CPU-baseline loop. Returns (edges_checked, neighbours, hits, nearest_idx)."""
    q = np.ascontiguousarray(queries, dtype=np.float64)
    nq = q.shape[0]
    nearest = np.empty(nq, dtype=np.int64)
    nn, nh = C.c_int64(), C.c_int64()
    e = lib().orc_extend_batch_spheres(tree.handle, spheres, m, _dp(q), nq, r, robot_radius,
                                       nearest.ctypes.data_as(c_int64_p), C.byref(nn), C.byref(nh))
    return e, nn.value, nh.value, nearest
