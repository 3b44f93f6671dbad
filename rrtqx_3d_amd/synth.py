"""Deterministic synthetic workloads for the hot path.

Nodes: i.i.d. uniform in [-50, 50]^3 (the law of randNodeDefault,
R/DRRT_Q.jl:600); Dubins adds t = 0 and theta ~ U[0, 2*pi) ([x y 0 theta],
R/dubinsExperimentsForPaper.jl:73-78).  Radius: R/rrtqx.jl:382 with the script
constants (R/experimentsForRRTQX.jl:132, R/dubinsExperimentsForPaper.jl:102).
Sphere obstacles: radius U[1, 3.5] (3.5 = environments/building2.txt).  Polygon
obstacles: the recipe of R/generate2DRandomDiscoverableObstacles.m:44-98.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

SEED = 20260101
WORLD = 50.0  # half-width


@dataclass
class Config:
    name: str
    n_nodes: int
    n_obstacles: int
    batch: int
    dim: int = 3
    dubins: bool = False


# BASELINE.json configs 2-5 (config 1 is the CPU plumbing case).  C5 is what BASELINE.json says: DubinsEdge in
# [x y t theta] with dynamic / discoverable obstacles at N = 500k (see dynamic_polygons and nodes_time below);
# C5s is the plain 3-D search at C5's node count.
CONFIGS = {
    "C2": Config("C2: 3D SimpleEdge N=10k M=32 B=1024", 10_000, 32, 1024),
    "C3": Config("C3: Dubins N=50k M=64 B=4096", 50_000, 64, 4096, dim=4, dubins=True),
    "C4": Config("C4: 3D SimpleEdge N=200k M=256 B=16384", 200_000, 256, 16384),
    "C5": Config("C5: DubinsEdge + time, dynamic discoverable obstacles, N=500k M=256 B=16384", 500_000, 256, 16384,
                 dim=4, dubins=True),
    "C5s": Config("C5s: 3D SimpleEdge search at N=500k M=256 B=16384", 500_000, 256, 16384),
}

# the Dubins-with-time scenes of R/dubinsExperimentsForPaper.jl:193-236: time spans [minTime, maxTime],
# velocities 5 .. 30, minTurningRadius 2, delta 20 / ballConstant 400 in the script (SURVEY 8d uses 10 / 100)
T_MIN, T_MAX = 10.0, 35.0
V_MIN, V_MAX = 5.0, 30.0
R_MIN_TIME = 2.0


def ball_radius(n: int, d: int, gamma: float = 80.0, delta: float = 8.0) -> float:
    """hyberBallRad = min(delta, ballConstant*((log(1+n)/n)^(1/d)))  (R/rrtqx.jl:382)"""
    return min(delta, gamma * ((math.log(1 + n) / n) ** (1.0 / d)))


def nodes(n: int, dim: int = 3, seed: int = SEED) -> np.ndarray:
    rng = np.random.default_rng(seed)
    if dim == 3:
        return rng.uniform(-WORLD, WORLD, size=(n, 3))
    xy = rng.uniform(-WORLD, WORLD, size=(n, 2))
    th = rng.uniform(0.0, 2.0 * math.pi, size=(n, 1))
    return np.concatenate([xy, np.zeros((n, 1)), th], axis=1)


def queries(b: int, dim: int = 3, seed: int = SEED + 1) -> np.ndarray:
    return nodes(b, dim, seed)


def nodes_time(n: int, seed: int = SEED) -> np.ndarray:
    """[x y t theta] with the time coordinate uniform over [T_MIN, T_MAX] (the bounds of the space,
    R/dubinsExperimentsForPaper.jl:209-210)."""
    p = nodes(n, 4, seed)
    rng = np.random.default_rng(seed + 9)
    p[:, 2] = rng.uniform(T_MIN, T_MAX, n)
    return p


def dynamic_polygons(m: int, seed: int = SEED + 4, moving_frac: float = 0.25, hidden_frac: float = 0.125):
    """BASELINE config 5's obstacle list: m polygons by the recipe of polygons(); the first
    moving_frac * m move in time (kinds 6 and 7 alternating) along piecewise-linear paths of 2 .. 12 rows
    (dx, dy, t), t ascending over [0, 45] (the path grammar of readTimeObstaclesFromfile,
    R/DRRT_Q.jl:1022-1061); the last hidden_frac * m are "discoverable": static polygons the robot has not
    seen yet (active = 0 until they appear, R/generate2DRandomDiscoverableObstacles.m).
    Returns (polys, kinds, paths, active, hidden_indices)."""
    rng = np.random.default_rng(seed)
    polys = polygons(m, seed)
    n_mov, n_hid = int(m * moving_frac), int(m * hidden_frac)
    kinds = np.full(m, 3, dtype=np.uint8)
    paths = [None] * m
    for i in range(n_mov):
        kinds[i] = 6 if i % 2 == 0 else 7
        rows = int(rng.integers(2, 13))
        t = np.sort(rng.uniform(0.0, 45.0, rows))
        t[0] = 0.0
        step = rng.normal(0.0, 6.0, (rows, 2))
        step[0] = 0.0
        paths[i] = np.concatenate([np.cumsum(step, axis=0), t[:, None]], axis=1)
    active = np.ones(m, dtype=np.uint8)
    hidden = np.arange(m - n_hid, m)
    active[hidden] = 0
    return polys, kinds, paths, active, hidden


def spheres(m: int, seed: int = SEED + 2) -> np.ndarray:
    rng = np.random.default_rng(seed)
    c = rng.uniform(-WORLD, WORLD, size=(m, 3))
    r = rng.uniform(1.0, 3.5, size=(m, 1))
    return np.concatenate([c, r], axis=1)


def polygons(m: int, seed: int = SEED + 3):
    """List of (P x 2) vertex arrays, P in {3, 4}; radius scaled so coverage stays
    roughly constant in m."""
    rng = np.random.default_rng(seed)
    scale = (100.0 / math.sqrt(max(m, 1))) / 10.0
    out = []
    for _ in range(m):
        max_radius = (5.0 + rng.random() * 10.0) * scale
        npts = 3 + int(math.floor(rng.random() * 2.0))
        centre = rng.uniform(-WORLD, WORLD, size=2)
        ang = np.sort(rng.random(npts) * 2.0 * math.pi)
        dist = np.sqrt(rng.random(npts)) * max_radius
        out.append(np.stack([dist * np.cos(ang) + centre[0], dist * np.sin(ang) + centre[1]], axis=1))
    return out


def candidate_edges(q: np.ndarray, node_pos: np.ndarray, offsets: np.ndarray, idx: np.ndarray):
    """Both directed edges of every (query, neighbour) pair, as extend() forms them
    (R/DRRT_Q.jl:1951, :2600).  Returns (p0, p1) with 2*len(idx) rows: out-edges
    first, then in-edges."""
    counts = np.diff(offsets)
    owner = np.repeat(np.arange(q.shape[0]), counts)
    a = q[owner]
    b = node_pos[idx]
    return np.concatenate([a, b], axis=0), np.concatenate([b, a], axis=0)
