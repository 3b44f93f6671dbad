"""Build recipe for librrtx_hip.so (gfx950 only, explicit hipcc, in-tree output).

    python -m rrtqx_3d_amd.build [--force] [--save-temps]

hipcc cross-compiles without a GPU.  -ffp-contract=off is mandatory: the
reference's arithmetic is unfused IEEE fp64 and neighbour sets / collision
booleans must match it bit for bit.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "librrtx_hip.so")
LIB_CLK = os.path.join(HERE, "librrtx_hip_clk.so")

SOURCES = ["rrtx_capi.hip", "kernels_nn.hip", "kernels_finish.hip", "kernels_nearest.hip", "kernels_slab.hip", "kernels_sweep.hip", "kernels_graph.hip", "kernels_collide.hip",
           "kernels_dubins.hip"]
HEADERS = ["rrtx_internal.hpp", "exact_math.hpp", "nn_device.hpp", "collide_device.hpp", os.path.join("..", "..", "include", "rrtx.h"),
           os.path.join("..", "..", "include", "rrtx_detmath.h")]

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]


def _newer(a: str, b: str) -> bool:
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _compile(src: str, force: bool, save_temps: bool, clocks: bool = False) -> str:
    obj = os.path.join(OBJDIR, src.replace(".hip", ".clk.o" if clocks else ".o"))
    srcp = os.path.join(CSRC, src)
    deps = [srcp] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    if force or any(_newer(d, obj) for d in deps):
        cmd = [HIPCC] + FLAGS + ["-c", srcp, "-o", obj]
        if clocks:
            cmd += ["-DRRTX_TILE_CLOCKS"]
        if save_temps:
            cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if save_temps and r.stderr:
            with open(obj + ".resource-usage.txt", "w") as f:
                f.write(r.stderr)
    return obj


def build(force: bool = False, save_temps: bool = False, clocks: bool = False) -> str:
    """clocks: the measuring build librrtx_hip_clk.so (phase clocks in the tile kernel, tools/tile_clocks.py);
    never loaded by the product, which always takes librrtx_hip.so."""
    os.makedirs(OBJDIR, exist_ok=True)
    lib = LIB_CLK if clocks else LIB
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, save_temps, clocks), SOURCES))
    if force or any(_newer(o, lib) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, clocks="--clocks" in sys.argv))
