"""The steady-state loop of bench.py alone (search on a fresh batch with the radius of the current n, then append of the
whole batch), for `rocprofv3 --kernel-trace --stats -- python3 tools/steady_trace.py`: which kernels a steady step costs."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

N, B, STEPS, RING = 200_000, 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 22, 8
dev = torch.device("cuda", 0)
pts = synth.nodes(N, 3)
ctx = Context(3, node_capacity=N + (STEPS + 2) * B)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
ctx.spheres_set(synth.spheres(256))
d_pts = torch.from_numpy(pts).to(dev)
ctx.nodes_append_dev(d_pts.data_ptr(), N)
q = [torch.from_numpy(synth.queries(B, 3, seed=synth.SEED + 1 + 17 * j)).to(dev) for j in range(RING)]
cap = 96 * B
off = torch.empty(B + 1, dtype=torch.int64, device=dev); idx = torch.empty(cap, dtype=torch.int32, device=dev)
cost = torch.empty(cap, dtype=torch.float64, device=dev); fl = torch.zeros(2 * cap + B, dtype=torch.uint8, device=dev)
ni = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.empty(B, dtype=torch.float64, device=dev)
need = torch.zeros(STEPS + 1, dtype=torch.int64, device=dev)


def step(i, slot):
    rr = synth.ball_radius(N + i * B, 3)
    ctx.extend_candidates_dev(q[i % RING].data_ptr(), B, rr, 0.5, off.data_ptr(), idx.data_ptr(), cost.data_ptr(),
                              fl.data_ptr(), fl.data_ptr() + cap, cap, need.data_ptr() + 8 * slot, ni.data_ptr(), nd.data_ptr(),
                              fl.data_ptr() + 2 * cap)
    ctx.nodes_append_dev(q[i % RING].data_ptr(), B)


step(0, STEPS)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(STEPS):
    step(i + 1, i)
torch.cuda.synchronize()
print("steady ms/step %.4f over %d steps" % (1e3 * (time.perf_counter() - t0) / STEPS, STEPS))
ctx.close()
