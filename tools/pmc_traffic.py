"""HBM-side bytes per launch of one kernel from the FETCH_SIZE / WRITE_SIZE passes of scripts_gpu_pmc.sh.

    python tools/pmc_traffic.py <dir with p3/p4/p5_per_kernel_avg.csv> "<kernel name prefix>" > traffic.json

FETCH_SIZE / WRITE_SIZE are in KB.  On gfx950 FETCH_SIZE tallies a 128-byte request of a wide (16 B per lane)
coalesced read as 64 bytes (MI355X_MICROARCH.md, HBM): the read side is doubled, as that guide prescribes; the
kernel's node reads are such loads.  WRITE_SIZE is exact for 16-byte-per-lane stores (the hit records).
"""
import csv
import hashlib
import json
import os
import sys

d, kern = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    h = hashlib.sha256()
    c = os.path.join(ROOT, "rrtqx_3d_amd", "csrc")
    for name in sorted(os.listdir(c)):
        if name.endswith((".hip", ".hpp")):
            with open(os.path.join(c, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def get(fname, counter):
    for r in csv.DictReader(open(os.path.join(d, fname))):
        if r["counter"] == counter and kern.replace(" ", "") in r["kernel"].replace(" ", ""):
            return float(r["avg_per_launch"]), int(r["launches"])
    return None, 0


fetch, n = get("p3_per_kernel_avg.csv", "FETCH_SIZE")
write, _ = get("p4_per_kernel_avg.csv", "WRITE_SIZE")
hit, _ = get("p5_per_kernel_avg.csv", "TCC_HIT_sum")
miss, _ = get("p5_per_kernel_avg.csv", "TCC_MISS_sum")
valu, _ = get("p1_per_kernel_avg.csv", "SQ_INSTS_VALU")
salu, _ = get("p1_per_kernel_avg.csv", "SQ_INSTS_SALU")
wcyc, _ = get("p1_per_kernel_avg.csv", "SQ_WAVE_CYCLES")
waves, _ = get("p1_per_kernel_avg.csv", "SQ_WAVES")
wait, _ = get("p2_per_kernel_avg.csv", "SQ_WAIT_ANY")
out = {
    "kernel": kern, "config": "C4", "launches_averaged": n,
    "SQ_INSTS_VALU_per_launch": valu, "SQ_INSTS_SALU_per_launch": salu, "SQ_WAVES_per_launch": waves,
    "SQ_WAVE_CYCLES_per_launch": wcyc, "SQ_WAIT_ANY_per_launch": wait,
    "wait_ratio": (wait / wcyc) if wait and wcyc else None,
    "FETCH_SIZE_bytes_per_launch": fetch * 1024, "WRITE_SIZE_bytes_per_launch": write * 1024,
    "traffic_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
    "l2_hit_rate": hit / (hit + miss) if hit is not None and hit + miss > 0 else None,
    "source_sha16": source_sha16(),
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KB -> bytes), averaged over the launches of the "
            "kernel in `python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras`; read side doubled as "
            "MI355X_MICROARCH.md prescribes for gfx950 (16 B/lane loads), write side = 16-byte hit records (exact) + "
            "8-byte screen entries (uncalibrated width)",
}
print(json.dumps(out, indent=1))
