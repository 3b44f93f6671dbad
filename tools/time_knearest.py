"""Wall time of rrtx_nn_knearest (host-buffer call, best of 3) on the BASELINE C4 tree (N = 200 k nodes):
the default path (k nearest taken from range-search lists) beside the exhaustive selection kernel."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import _capi, synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402


def main():
    N = 200000
    pts = synth.nodes(N, 3)
    out = []
    with Context(3, node_capacity=N) as ctx:
        ctx.nodes_append(pts)
        for B, k in ((1024, 16), (4096, 16), (16384, 16), (4096, 256)):
            Q = synth.queries(B, 3)
            row = {"n_nodes": N, "queries": B, "k": k}
            for name, lists in (("lists", 1), ("exhaustive", 0)):
                ctx.set_option(_capi.RRTX_OPT_KNN_LISTS, lists)
                ctx.nn_knearest(Q, k)                      # same shape once: workspace allocations
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    ctx.nn_knearest(Q, k)
                    best = min(best, time.perf_counter() - t0)
                row["ms_" + name] = round(best * 1e3, 3)
                row["queries_per_s_" + name] = round(B / best, 1)
            out.append(row)
            print(json.dumps(out[-1]), flush=True)


if __name__ == "__main__":
    sys.exit(main())
