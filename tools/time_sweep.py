"""Times rrtx_obstacle_sweep at C4 scale: 200k nodes, ~50 out-edges per node mirrored on the device."""
import json, time
import numpy as np
from rrtqx_3d_amd import synth
from rrtqx_3d_amd.context import Context

n, per = 200_000, 50
rng = np.random.default_rng(1)
pts = synth.nodes(n, 3)
es = np.repeat(np.arange(n, dtype=np.int32), per)
ee = ((es + rng.integers(1, 2000, len(es))) % n).astype(np.int32)
sph = synth.spheres(256)
with Context(3, node_capacity=n) as ctx:
    ctx.nodes_append(pts)
    ctx.spheres_set(sph)
    t0 = time.perf_counter()
    ctx.graph_edges_append(es, ee)
    t_up = time.perf_counter() - t0
    rr, delta = 0.5, 8.0
    ids = ctx.obstacle_sweep(0, rr + delta + sph[0, 3], rr, cap=1 << 20)
    ctx.profile(2)
    t0 = time.perf_counter()
    reps = 20
    for j in range(reps):
        ids = ctx.obstacle_sweep(j, rr + delta + sph[j, 3], rr, cap=1 << 20)
    wall = (time.perf_counter() - t0) / reps
    st = ctx.stats()
    print(json.dumps({"case": "obstacle_sweep, 200k nodes, 10M mirrored edges", "upload_ms": round(t_up * 1e3, 2),
                      "wall_ms_per_sweep": round(wall * 1e3, 4), "device_ms_per_sweep": round(st.ms_edges / reps, 4),
                      "colliding_edges_last": int(len(ids))}))
