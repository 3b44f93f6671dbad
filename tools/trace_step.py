#!/usr/bin/env python3
"""Prints the kernel sequence of one bench step with start offsets and gaps from a rocprofv3
kernel-trace CSV (tools/trace_step.py <kernel_trace.csv>)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# take the last complete step: from the last-but-one nn_pack kernel to the last one
packs = [i for i, r in enumerate(rows) if "nn_pack_kernel" in r["Kernel_Name"]]
k = len(packs) // 2
a, b = packs[k], packs[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = None
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:6.1f} us  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"step period {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
