"""Does replaying the fused extend() step from a captured HIP graph shorten the gaps between its four dependent
launches?  Captures RING (8) consecutive steps (the calls alternate between two per-call state records, so an even
number of calls per graph keeps the alternation) through torch's stream capture and compares direct calls with
graph replays.  GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

cfg = synth.CONFIGS["C4"]
N, B, RING = cfg.n_nodes, cfg.batch, 8
dev = torch.device("cuda:0")
r = synth.ball_radius(N, 3)
cap = 96 * B
ctx = Context(3, device=0, node_capacity=N)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    ctx.set_stream(side.cuda_stream)
    ctx.spheres_set(synth.spheres(cfg.n_obstacles))
    d_pts = torch.from_numpy(synth.nodes(N, 3)).to(dev)
    ctx.nodes_append_dev(d_pts.data_ptr(), N)
    qs = [torch.from_numpy(synth.queries(B, 3, seed=synth.SEED + 1 + 17 * j)).to(dev) for j in range(RING)]
    off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
    idx = torch.zeros(cap, dtype=torch.int32, device=dev)
    cost = torch.zeros(cap, dtype=torch.float64, device=dev)
    ho = torch.zeros(cap, dtype=torch.uint8, device=dev)
    hi = torch.zeros(cap, dtype=torch.uint8, device=dev)
    need = torch.zeros(RING, dtype=torch.int64, device=dev)
    nidx = torch.zeros(B, dtype=torch.int32, device=dev)
    nd = torch.zeros(B, dtype=torch.float64, device=dev)
    uns = torch.zeros(B, dtype=torch.uint8, device=dev)

    def step(j):
        ctx.extend_candidates_dev(qs[j].data_ptr(), B, r, 0.5, off.data_ptr(), idx.data_ptr(), cost.data_ptr(),
                                  ho.data_ptr(), hi.data_ptr(), cap, need.data_ptr() + 8 * j, nidx.data_ptr(),
                                  nd.data_ptr(), uns.data_ptr())

    for _ in range(3):
        for j in range(RING):
            step(j)
    side.synchronize()
    ref_need = need.clone()
    ref_idx = idx.clone()
    t0 = time.perf_counter()
    for _ in range(10):
        for j in range(RING):
            step(j)
    side.synchronize()
    t_direct = (time.perf_counter() - t0) / (10 * RING)
    print("direct calls : %.4f ms per step" % (1e3 * t_direct))
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=side):
            for j in range(RING):
                step(j)
    except Exception as e:   # noqa: BLE001
        print("capture failed:", repr(e)[:300])
        sys.exit(0)
    for _ in range(3):
        g.replay()
    side.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    side.synchronize()
    t_graph = (time.perf_counter() - t0) / (10 * RING)
    print("graph replays: %.4f ms per step" % (1e3 * t_graph))
    print("same results :", bool(torch.equal(need, ref_need)) and bool(torch.equal(idx[: int(need[RING - 1])], ref_idx[: int(need[RING - 1])])))
ctx.close()
