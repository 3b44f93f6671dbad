"""Host-side cost of issuing one bench step (no sync inside the loop) vs the GPU time per step."""
import time, sys
import numpy as np, torch
from rrtqx_3d_amd import synth
from rrtqx_3d_amd.context import Context
cfg = synth.CONFIGS["C4"]
N, M, B = cfg.n_nodes, cfg.n_obstacles, cfg.batch
r = synth.ball_radius(N, 3)
dev = torch.device("cuda", 0)
ctx = Context(3)
ctx.nodes_append(synth.nodes(N, 3)); ctx.spheres_set(synth.spheres(M))
Q = torch.from_numpy(synth.queries(B, 3)).to(dev)
cap = 96 * B
off = torch.empty(B + 1, dtype=torch.int64, device=dev); idx = torch.empty(cap, dtype=torch.int32, device=dev)
cost = torch.empty(cap, dtype=torch.float64, device=dev); ho = torch.empty(cap, dtype=torch.uint8, device=dev)
hi = torch.empty(cap, dtype=torch.uint8, device=dev); need = torch.empty(1, dtype=torch.int64, device=dev)
ni = torch.empty(B, dtype=torch.int32, device=dev); nd = torch.empty(B, dtype=torch.float64, device=dev)
un = torch.empty(B, dtype=torch.uint8, device=dev)
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
def step():
    ctx.extend_candidates_dev(Q.data_ptr(), B, r, 0.5, off.data_ptr(), idx.data_ptr(), cost.data_ptr(), ho.data_ptr(),
                              hi.data_ptr(), cap, need.data_ptr(), ni.data_ptr(), nd.data_ptr(), un.data_ptr())
for _ in range(20): step()
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n): step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f"issue {1e6*t_issue/n:.1f} us/step, total {1e6*t_all/n:.1f} us/step")
# graph capture of one step
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ctx.set_stream(s.cuda_stream)
    step(); s.synchronize()
    try:
        with torch.cuda.graph(g, stream=s):
            step()
        for _ in range(20): g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): g.replay()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"graph: issue {1e6*t_issue/n:.1f} us/step, total {1e6*t_all/n:.1f} us/step")
    except Exception as e:
        print("graph capture failed:", repr(e)[:300])
ctx.close()
