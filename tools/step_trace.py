"""A few bench steps (extend_candidates_dev on resident inputs, config C4) for rocprofv3 --kernel-trace."""
import sys
import numpy as np
import torch
from rrtqx_3d_amd import synth
from rrtqx_3d_amd.context import Context

cfg = synth.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C4"]
N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
r = synth.ball_radius(N, 3)
dev = torch.device("cuda", 0)
ctx = Context(3)
ctx.nodes_append(synth.nodes(N, 3))
ctx.spheres_set(synth.spheres(M))
Q = torch.from_numpy(synth.queries(B, 3)).to(dev)
cap = 96 * B
off = torch.empty(B + 1, dtype=torch.int64, device=dev)
idx = torch.empty(cap, dtype=torch.int32, device=dev)
cost = torch.empty(cap, dtype=torch.float64, device=dev)
ho = torch.empty(cap, dtype=torch.uint8, device=dev)
hi = torch.empty(cap, dtype=torch.uint8, device=dev)
need = torch.empty(1, dtype=torch.int64, device=dev)
ni = torch.empty(B, dtype=torch.int32, device=dev)
nd = torch.empty(B, dtype=torch.float64, device=dev)
un = torch.empty(B, dtype=torch.uint8, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    ctx.extend_candidates_dev(Q.data_ptr(), B, r, 0.5, off.data_ptr(), idx.data_ptr(), cost.data_ptr(), ho.data_ptr(),
                              hi.data_ptr(), cap, need.data_ptr(), ni.data_ptr(), nd.data_ptr(), un.data_ptr())
torch.cuda.synchronize()
print("neighbours", int(need.item()))
ctx.close()
