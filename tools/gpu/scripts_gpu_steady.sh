#!/bin/bash
# kernel stats of the steady-state loop alone (bench.py --steady-only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/steady
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --no-cpu-baseline > $out/bench.json 2> $out/rocprof.err
echo "rc=$?"
cp $out/trace/*/*_kernel_stats.csv $out/kernel_stats.csv
rm -rf $out/trace
python3 - <<PY
import csv, re
rows = []
for r in csv.reader(open("$out/kernel_stats.csv")):
    if r[0] == "Name": continue
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r[0])
    rows.append((float(r[2]) / 1000, int(r[1]), float(r[3]) / 1000, m.group(1) if m else r[0][:40]))
rows.sort(reverse=True)
for tot, calls, avg, name in rows[:22]:
    print("%-44s calls %5d total %9.1f us avg %8.2f" % (name[:44], calls, tot, avg))
PY
