#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r3f
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --obstacles polygons --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $out/b.json 2> $out/b.err
f=$(ls $out/trace/*/*_kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
last = rows[-60:]
for r in last:
    n = r["Kernel_Name"]
    n = n[n.find("::", 20) + 2:][:40] if "rrtx" in n else n[:40]
    print("%-42s q=%s start %9.1f us  dur %7.1f us" % (n, r.get("Queue_Id"), (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
rm -rf $out/trace
