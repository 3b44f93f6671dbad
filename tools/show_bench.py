"""one-line summary of a bench.py JSON line (any config)"""
import json
import sys

for path in sys.argv[1:]:
    with open(path) as f:
        line = [l for l in f if l.startswith("{")][-1]
    d = json.loads(line)
    print("%s: %s edges/s %.4g  ms/step %.4f  kernels %s" % (path, d["config"]["workload"][:12], d["value"], d["ms_per_step"],
                                                           {k: round(v, 4) for k, v in d["kernel_ms"].items()}))
