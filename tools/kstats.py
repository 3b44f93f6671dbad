"""print the rrtx kernels of a rocprofv3 *_kernel_stats.csv: calls, average / min duration"""
import csv
import re
import sys

for path in sys.argv[1:]:
    print(path)
    for r in csv.reader(open(path)):
        if r[0] == "Name" or "rrtx" not in r[0]:
            continue
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", r[0])
        n = m.group(1) if m else r[0][:40]
        print("   %-42s calls %5s avg %8.2f us  min %8.2f" % (n[:42], r[1], float(r[3]) / 1000, float(r[5]) / 1000))
