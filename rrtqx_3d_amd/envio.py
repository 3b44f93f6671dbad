"""Environment files of the reference: the two text grammars
the planner loads obstacles from, so fixtures load without Julia.

  spheres   (R/DRRT_Q.jl:901-947, readDiscoverable3DObstaclesFromfile,
             e.g. environments/building2.txt):
      count
      x, y, z          \\
      radius            > per obstacle
      behaviour        /   0 normal | -1 vanishes when sensed | 1 appears when sensed

  polygons  (R/DRRT_Q.jl:853-899, readDiscoverablecObstaclesFromfile,
             e.g. environments/rand_Static.txt):
      count
      nverts           \\
      x, y   (x nverts)  > per obstacle
      behaviour        /

  moving polygons  (R/DRRT_Q.jl:1022-1061 = R/DRRT.jl:877-916, readTimeObstaclesFromfile, kind 6;
             readDynamicTimeObstaclesFromfile :1067-1109 reads the same grammar into kind 7's
             unknownPath; e.g. environments/rand_StaticTime.txt):
      count
      nverts           \\
      x, y   (x nverts)   |
      speed               > per obstacle
      npath               |
      dx, dy, t (x npath) /   offsets from the polygon's ctor position vs time (ascending t)

Obstacles are pushed to the FRONT of CSpace.obstacles (addObsToCSpace ->
listPush, R/list.jl:53-58), so list order is the reverse of file order;
`list_order()` applies that.  Behaviour 1 ("appears") starts with
obstacleUnused = true (:932-935), i.e. inactive until the robot senses it.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np


def str2array(s: str) -> np.ndarray:
    """R/DRRT_Q.jl:171-210: comma separated floats on one line."""
    return np.array([float(t) for t in s.strip().split(",") if t.strip() != ""], dtype=np.float64)


def jl_float_str(x: float) -> str:
    """A Float64 the way Julia 1.0's writedlm / show prints it (base/grisu/grisu.jl `_show`): shortest
    round-trip digits; plain notation while the decimal point position pt satisfies -4 < pt <= 6, else
    d.ddde<exp>; always at least one digit after the point; Inf / -Inf / NaN."""
    x = float(x)
    if x != x:
        return "NaN"
    if x in (float("inf"), float("-inf")):
        return "Inf" if x > 0 else "-Inf"
    sign = "-" if (x < 0 or (x == 0 and str(x).startswith("-"))) else ""
    r = repr(abs(x))
    if "e" in r:
        mant, exp = r.split("e")
        exp = int(exp)
    else:
        mant, exp = r, 0
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    pt = len(ip.lstrip("0")) + exp if ip.strip("0") else exp - (len(fp) - len(fp.lstrip("0")))
    digits = digits.rstrip("0")
    if not digits:
        return sign + "0.0"
    if pt <= -4 or pt > 6:
        return sign + digits[0] + "." + (digits[1:] or "0") + "e" + str(pt - 1)
    if pt <= 0:
        return sign + "0." + "0" * (-pt) + digits
    if pt >= len(digits):
        return sign + digits + "0" * (pt - len(digits)) + ".0"
    return sign + digits[:pt] + "." + digits[pt:]


def writedlm_row(f, values):
    """One row of writedlm(fptr, row, ',')."""
    f.write(",".join(jl_float_str(v) for v in values) + "\n")


@dataclass
class SphereEnv:
    cxyzr: np.ndarray        # m x 4, FILE order
    behaviour: np.ndarray    # m, int

    def active(self) -> np.ndarray:
        """obstacleUnused == false at load time (behaviour 0 and -1)."""
        return (self.behaviour != 1).astype(np.uint8)

    def list_order(self):
        """(cxyzr, active) in CSpace list order (front first)."""
        return self.cxyzr[::-1].copy(), self.active()[::-1].copy()


@dataclass
class PolygonEnv:
    polygons: List[np.ndarray]   # each P x 2, FILE order
    behaviour: np.ndarray

    def active(self) -> np.ndarray:
        return (self.behaviour != 1).astype(np.uint8)

    def list_order(self):
        return [p.copy() for p in self.polygons[::-1]], self.active()[::-1].copy()


@dataclass
class TimeObstacleEnv:
    polygons: List[np.ndarray]   # each P x 2 (originalPolygon), FILE order
    speed: np.ndarray            # obsSpeed per obstacle (Obstacle.velocity)
    paths: List[np.ndarray]      # each M x 3 rows of (dx, dy, t)

    def list_order(self):
        return [p.copy() for p in self.polygons[::-1]], [p.copy() for p in self.paths[::-1]]


def read_sphere_obstacles(path: str) -> SphereEnv:
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip() != ""]
    n = int(lines[0])
    rows, beh = [], []
    p = 1
    for _ in range(n):
        c = str2array(lines[p])
        if c.shape[0] != 3:
            raise ValueError(f"{path}: expected 'x, y, z' on line {p + 1}")
        r = float(lines[p + 1])
        b = int(lines[p + 2])
        if b not in (0, -1, 1):
            raise ValueError("unknown behavoiur type")   # the reference's error text, R/DRRT_Q.jl:937
        rows.append([c[0], c[1], c[2], r])
        beh.append(b)
        p += 3
    return SphereEnv(np.array(rows, dtype=np.float64).reshape(-1, 4), np.array(beh, dtype=np.int32))


def read_polygon_obstacles(path: str) -> PolygonEnv:
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip() != ""]
    n = int(lines[0])
    polys, beh = [], []
    p = 1
    for _ in range(n):
        nv = int(lines[p])
        v = np.stack([str2array(lines[p + 1 + k]) for k in range(nv)], axis=0)
        if v.shape[1] != 2:
            raise ValueError(f"{path}: expected 'x, y' vertex rows")
        b = int(lines[p + 1 + nv])
        if b not in (0, -1, 1):
            raise ValueError("unknown behavoiur type")
        polys.append(v)
        beh.append(b)
        p += nv + 2
    return PolygonEnv(polys, np.array(beh, dtype=np.int32))


def read_time_obstacles(path: str) -> TimeObstacleEnv:
    with open(path) as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip() != ""]
    n = int(lines[0])
    polys, speed, paths = [], [], []
    p = 1
    for _ in range(n):
        nv = int(lines[p])
        v = np.stack([str2array(lines[p + 1 + k]) for k in range(nv)], axis=0)
        if v.shape[1] != 2:
            raise ValueError(f"{path}: expected 'x, y' vertex rows")
        p += 1 + nv
        speed.append(float(lines[p]))
        m = int(lines[p + 1])
        mp = np.stack([str2array(lines[p + 2 + k]) for k in range(m)], axis=0)
        if mp.shape[1] != 3:
            raise ValueError(f"{path}: expected 'dx, dy, t' path rows")
        p += 2 + m
        polys.append(v)
        paths.append(mp)
    return TimeObstacleEnv(polys, np.array(speed, dtype=np.float64), paths)


def write_time_obstacles(path: str, env: TimeObstacleEnv):
    with open(path, "w") as f:
        f.write(f"{len(env.polygons)}\n")
        for v, sp, mp in zip(env.polygons, env.speed, env.paths):
            f.write(f"{v.shape[0]}\n")
            for x, y in v:
                f.write(f"{x:f}, {y:f}\n")
            f.write(f"{sp:f}\n{mp.shape[0]}\n")
            for dx, dy, t in mp:
                f.write(f"{dx:f}, {dy:f}, {t:f}\n")


def write_sphere_obstacles(path: str, env: SphereEnv):
    with open(path, "w") as f:
        f.write(f"{env.cxyzr.shape[0]}\n")
        for (x, y, z, r), b in zip(env.cxyzr, env.behaviour):
            f.write(f"{float(x)!r}, {float(y)!r}, {float(z)!r}\n{float(r)!r}\n{int(b)}\n")


def write_polygon_obstacles(path: str, env: PolygonEnv):
    with open(path, "w") as f:
        f.write(f"{len(env.polygons)}\n")
        for v, b in zip(env.polygons, env.behaviour):
            f.write(f"{v.shape[0]}\n")
            for x, y in v:
                f.write(f"{x:f}, {y:f}\n")       # generate2DRandomDiscoverableObstacles.m:88 uses %f
            f.write(f"{int(b)}\n")
