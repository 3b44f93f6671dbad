"""ctypes binding of librrtx_hip.so (include/rrtx.h).

This is the same boundary a Julia `ccall` shim binds (julia/RRTXHip.jl); there is
no CPU fallback: if the HIP library is missing or no GPU is present, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librrtx_hip.so")

RRTX_OK = 0
RRTX_E_INVALID = -1
RRTX_E_CAPACITY = -2
RRTX_E_DEVICE = -3
RRTX_E_NOMEM = -4
RRTX_E_STATE = -5
RRTX_OPT_NN_FILTER = 1
RRTX_OPT_SCAN_BLOCKS = 2
RRTX_OPT_SCAN_TILE_Q = 3
RRTX_OPT_SCAN_ITEMS = 4
RRTX_OPT_NN_CULL = 5
RRTX_OPT_PROFILE_EVERY = 6
RRTX_OPT_KNN_LISTS = 7
RRTX_OPT_EXTEND_OBSTACLES = 8
RRTX_OPT_NEAREST_REC_CAP = 9
RRTX_OPT_BUCKET_MULT = 10
RRTX_OPT_TUNE = 11
RRTX_OPT_SPACE_HAS_TIME = 12
RRTX_OPT_ROOT_RULE = 13

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_int64_p = C.POINTER(C.c_int64)
c_uint8_p = C.POINTER(C.c_uint8)


class RrtxError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"rrtx error {code}: {msg}")
        self.code = code


class Stats(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int64), ("dim", C.c_int32), ("n_spheres", C.c_int32), ("n_polygons", C.c_int32),
        ("n_wraps", C.c_int32),
        ("ms_nn_scan", C.c_double), ("launches_nn_scan", C.c_int64),
        ("ms_nn_finish", C.c_double), ("launches_nn_finish", C.c_int64),
        ("ms_nn_nearest", C.c_double), ("launches_nn_nearest", C.c_int64),
        ("ms_edges", C.c_double), ("launches_edges", C.c_int64),
        ("ms_points", C.c_double), ("launches_points", C.c_int64),
        ("ms_dubins", C.c_double), ("launches_dubins", C.c_int64),
        ("last_pairs", C.c_int64), ("last_neighbors", C.c_int64),
        ("last_tile_q", C.c_int32), ("last_scan_units", C.c_int32),
        ("ms_dubins_steer", C.c_double), ("launches_dubins_steer", C.c_int64),
        ("last_sweep_candidates", C.c_int64),
    ]


# every symbol include/rrtx.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ("rrtx_create", C.c_int, [C.POINTER(_VP), C.c_int, C.c_int, C.c_int64]),
    ("rrtx_destroy", C.c_int, [_VP]),
    ("rrtx_last_error", C.c_char_p, [_VP]),
    ("rrtx_create_error", C.c_char_p, []),
    ("rrtx_sq_thresholds", C.c_int, [C.c_double, c_double_p, c_double_p]),
    ("rrtx_set_stream", C.c_int, [_VP, _VP]),
    ("rrtx_get_stream", _VP, [_VP]),
    ("rrtx_sync", C.c_int, [_VP]),
    ("rrtx_profile", C.c_int, [_VP, C.c_int]),
    ("rrtx_stats", C.c_int, [_VP, C.POINTER(Stats)]),
    ("rrtx_set_option", C.c_int, [_VP, C.c_int, C.c_int64]),
    ("rrtx_get_option", C.c_int, [_VP, C.c_int, c_int64_p]),
    ("rrtx_host_register", C.c_int, [_VP, _VP, C.c_size_t]),
    ("rrtx_host_unregister", C.c_int, [_VP, _VP]),
    ("rrtx_nodes_append", C.c_int, [_VP, _VP, C.c_int64, c_int64_p]),
    ("rrtx_nodes_count", C.c_int64, [_VP]),
    ("rrtx_nodes_append_dev", C.c_int, [_VP, _VP, C.c_int64]),
    ("rrtx_set_wrap", C.c_int, [_VP, C.c_int, C.c_double]),
    ("rrtx_spheres_set", C.c_int, [_VP, _VP, _VP, C.c_int]),
    ("rrtx_polygons_set", C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, C.c_int]),
    ("rrtx_polygon_paths_set", C.c_int, [_VP, _VP, _VP, C.c_int]),
    ("rrtx_obstacle_update", C.c_int, [_VP, C.c_int, C.c_double, C.c_uint8]),
    ("rrtx_nn_nearest", C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    ("rrtx_nn_knearest", C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    ("rrtx_nn_radius", C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP, C.c_int64, c_int64_p]),
    ("rrtx_edges_check", C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int64, C.c_double, C.c_int, _VP, _VP]),
    ("rrtx_edges_check_idx", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, C.c_int, _VP, _VP, _VP]),
    ("rrtx_graph_edges_append", C.c_int, [_VP, _VP, _VP, C.c_int64, c_int64_p]),
    ("rrtx_graph_edges_count", C.c_int64, [_VP]),
    ("rrtx_graph_edges_clear", C.c_int, [_VP]),
    ("rrtx_obstacle_sweep", C.c_int, [_VP, C.c_int, C.c_double, C.c_double, _VP, C.c_int64, c_int64_p]),
    ("rrtx_points_check", C.c_int, [_VP, C.c_int, _VP, C.c_int64, C.c_double, C.c_int, _VP, _VP]),
    ("rrtx_simple_steer", C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP]),
    ("rrtx_dubins_steer", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, _VP, _VP]),
    ("rrtx_graph_edges_set_dist", C.c_int, [_VP, C.c_int64, _VP, C.c_int64]),
    ("rrtx_graph_edges_block", C.c_int, [_VP, _VP, C.c_int64]),
    ("rrtx_graph_cost_to_root", C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    ("rrtx_graph_cost_to_root_dev", C.c_int, [_VP, C.c_int, _VP, _VP]),
    ("rrtx_graph_cost_update", C.c_int, [_VP, C.c_int, _VP, _VP, _VP]),
    ("rrtx_graph_cost_update_dev", C.c_int, [_VP, C.c_int, _VP, _VP]),
    ("rrtx_set_dubins_velocity", C.c_int, [_VP, C.c_double, C.c_double]),
    ("rrtx_dubins_steer_full", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, _VP, _VP, _VP, _VP, _VP]),
    ("rrtx_dubins_edges_check", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, C.c_double, _VP, _VP, _VP, _VP]),
    ("rrtx_dubins_trajectory", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, _VP, _VP, C.c_int, C.c_int64, c_int64_p]),
    ("rrtx_detmath_eval", C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int64, _VP]),
    ("rrtx_obstacle_sweep_polygon", C.c_int, [_VP, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int, _VP, C.c_int64, c_int64_p]),
    ("rrtx_dubins_edges_check_obstacle", C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, C.c_double, C.c_int, _VP]),
    ("rrtx_extend_candidates", C.c_int, [_VP, _VP, C.c_int, C.c_double, C.c_double, _VP, _VP, _VP, _VP, _VP,
                                         C.c_int64, c_int64_p, _VP, _VP, _VP]),
    ("rrtx_extend_candidates_dubins", C.c_int, [_VP, _VP, C.c_int, C.c_double, C.c_double, C.c_double, _VP, _VP, _VP, _VP,
                                                _VP, _VP, _VP, _VP, _VP, C.c_int64, c_int64_p, _VP, _VP, _VP]),
    ("rrtx_extend_candidates_dubins_dev", C.c_int, [_VP, _VP, C.c_int, C.c_double, C.c_double, C.c_double, _VP, _VP, _VP,
                                                    _VP, _VP, _VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, _VP, _VP]),
    ("rrtx_nn_nearest_dev", C.c_int, [_VP, _VP, C.c_int, _VP, _VP]),
    ("rrtx_nn_knearest_dev", C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, _VP, _VP]),
    ("rrtx_nn_radius_dev", C.c_int, [_VP, _VP, C.c_double, C.c_int, _VP, _VP, _VP, C.c_int64, _VP]),
    ("rrtx_edges_check_dev", C.c_int, [_VP, C.c_int, _VP, _VP, C.c_int64, C.c_double, C.c_int, C.c_int, C.c_int,
                                       _VP, _VP]),
    ("rrtx_points_check_dev", C.c_int, [_VP, C.c_int, _VP, C.c_int64, C.c_double, C.c_int, _VP, _VP]),
    ("rrtx_extend_candidates_dev", C.c_int, [_VP, _VP, C.c_int, C.c_double, C.c_double, _VP, _VP, _VP, _VP, _VP,
                                             C.c_int64, _VP, _VP, _VP, _VP]),
    ("rrtx_pack_hits_dev", C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _VP]),
]

_lib = None
_runtime = None


def mapped_hip_runtimes():
    """Real paths of every libamdhip64 image mapped into this process (/proc/self/maps)."""
    found = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if "libamdhip64.so" in os.path.basename(path):
                    rp = os.path.realpath(path)
                    if rp not in found:
                        found.append(rp)
    except OSError:
        pass
    return found


def hip_runtime() -> dict:
    """Which HIP runtime serves this process: librrtx_hip.so is built against the system ROCm while a
    PyTorch wheel bundles its own libamdhip64 under the same SONAME, so whichever image is loaded first
    serves both (stream handles and device pointers cross between torch and this library).  That is
    fine as long as it is ONE image; two different images in one process is refused."""
    load()
    _pin_runtime()          # look again: torch may have been imported since
    return dict(_runtime)


_verified_with_torch = False


def verify_runtime():
    """Called when a context is created: once torch is in the process, make sure it did not bring a
    second runtime image with it (it does when it is imported AFTER this library)."""
    global _verified_with_torch
    if _verified_with_torch or "torch" not in sys.modules:
        return
    _pin_runtime()
    _verified_with_torch = True


def _pin_runtime():
    global _runtime
    images = mapped_hip_runtimes()
    if len(images) > 1:
        raise RuntimeError("two different HIP runtimes are mapped into this process: " + ", ".join(images) +
                           " -- device pointers and streams would cross between them; import torch before "
                           "rrtqx_3d_amd (or not at all) so that one image serves both")
    info = {"path": images[0] if images else None, "version": None}
    if images:
        try:
            rt = C.CDLL(images[0])
            v = C.c_int(0)
            if rt.hipRuntimeGetVersion(C.byref(v)) == 0:
                info["version"] = int(v.value)
        except (OSError, AttributeError):
            pass
    _runtime = info


def load() -> C.CDLL:
    """Load librrtx_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m rrtqx_3d_amd.build` "
            "(the HIP extension is the only implementation; there is no CPU fallback)")
    # A process that also uses PyTorch holds two ROCm runtimes (the wheel bundles its own; this library
    # links the system one) and the bundled one has to be initialised first, or torch later reports
    # "No HIP GPUs are available".  If torch is already imported, bring its runtime up now.
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.init()
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    _pin_runtime()
    return L


def _ptr(a):
    """numpy array -> void* (None stays NULL)."""
    if a is None:
        return None
    return a.ctypes.data


def f64(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a
