"""rrtx_nn_knearest under rocprofv3: a few list-path calls at one shape (kernel split)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from rrtqx_3d_amd import synth  # noqa: E402
from rrtqx_3d_amd.context import Context  # noqa: E402

N, B, k = 200000, int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 16
with Context(3, node_capacity=N) as ctx:
    ctx.nodes_append(synth.nodes(N, 3))
    Q = synth.queries(B, 3)
    for _ in range(4):
        ctx.nn_knearest(Q, k)
