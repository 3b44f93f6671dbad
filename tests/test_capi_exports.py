"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/rrtx.h declares.  No compute calls (no GPU here)."""
import os
import re

from rrtqx_3d_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rrtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rrtx_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_survey_boundary():
    syms = declared_symbols()
    for must in ["rrtx_create", "rrtx_destroy", "rrtx_nodes_append", "rrtx_set_wrap", "rrtx_spheres_set",
                 "rrtx_polygons_set", "rrtx_obstacle_update", "rrtx_nn_nearest", "rrtx_nn_radius",
                 "rrtx_edges_check", "rrtx_points_check", "rrtx_dubins_steer", "rrtx_last_error", "rrtx_stats"]:
        assert must in syms


def test_library_exports_every_declared_symbol(hip_lib):
    for name in declared_symbols():
        assert hasattr(hip_lib, name), f"{name} declared in include/rrtx.h but not exported"


def test_binding_table_matches_header(hip_lib):
    bound = sorted(n for n, _, _ in _capi.SYMBOLS)
    assert bound == declared_symbols()


def test_signatures_are_plain_c(hip_lib):
    # no torch / C++ types may appear in the boundary
    text = open(os.path.join(ROOT, "include", "rrtx.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert "torch" not in text and "std::" not in text and "at::" not in text
    assert 'extern "C"' in text


def test_create_fails_loudly_without_gpu(hip_lib):
    """On a box without a GPU the product path must fail, not fall back."""
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        return
    h = C.c_void_p()
    rc = hip_lib.rrtx_create(C.byref(h), 3, 0, 1024)
    assert rc == _capi.RRTX_E_DEVICE
    assert b"HIP device" in hip_lib.rrtx_create_error()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rrtqx_3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "rrtx_oracle" not in src and "from oracle" not in src and "import oracle" not in src, f
