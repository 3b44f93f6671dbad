"""Device time of rrtx_nn_knearest at BASELINE config C4 (N = 200 k nodes), host-buffer call."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402


def main():
    N = 200000
    pts = synth.nodes(N, 3)
    out = []
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        for B, k in ((1024, 16), (4096, 16), (16384, 16), (4096, 256)):
            Q = synth.queries(B, 3)
            ctx.nn_knearest(Q[:64], k)
            t0 = time.perf_counter()
            ctx.nn_knearest(Q, k)
            dt = time.perf_counter() - t0
            out.append({"n_nodes": N, "queries": B, "k": k, "ms_host_call": round(dt * 1e3, 3),
                        "queries_per_s": round(B / dt, 1)})
            print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    sys.exit(main())
