"""D2H rates of this box for the sizes the host-pointer path moves (tools/gpu/r3_host.sh context)."""
import time
import torch
dev = torch.device("cuda", 0)
for mb in (0.4, 1.65, 3.3, 6.1, 24.0):
    n = int(mb * 1e6)
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    hp = torch.empty(n, dtype=torch.uint8)
    for name, dst in (("pinned", h), ("pageable", hp)):
        dst.copy_(d); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dst.copy_(d, non_blocking=True)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"{mb:5.2f} MB {name:8s} {dt * 1e6:8.1f} us  {n / dt / 1e9:6.1f} GB/s")
    t0 = time.perf_counter()
    for _ in range(20):
        hp.copy_(h)
    dt = (time.perf_counter() - t0) / 20
    print(f"{mb:5.2f} MB host memcpy {dt * 1e6:8.1f} us  {n / dt / 1e9:6.1f} GB/s")
