"""Average rocprofv3 --pmc counter values per (kernel, counter) over the launches of one pass.

    python tools/pmc_summary.py <pass_dir> > pN_per_kernel_avg.csv

<pass_dir> is the -d directory of one rocprofv3 --pmc run (any depth; *_counter_collection.csv).
"""
import csv, glob, os, sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for path in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            k = (row["Kernel_Name"][:90], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "avg_per_launch", "launches"])
for (kern, ctr), (tot, n) in sorted(acc.items()):
    w.writerow([kern, ctr, round(tot / n, 1), n])
