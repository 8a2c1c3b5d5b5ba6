"""Per-kernel averages of one rocprofv3 --pmc pass: reads <dir>/**/*counter_collection.csv, prints
kernel, counter, dispatches, mean value per dispatch (CSV on stdout)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0, 0.0])
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            key = (row["Kernel_Name"], row["Counter_Name"])
            acc[key][0] += 1
            acc[key][1] += float(row["Counter_Value"])
w = csv.writer(sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch", "total"])
for (k, c), (n, s) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    w.writerow([k[:160], c, n, s / n, s])
