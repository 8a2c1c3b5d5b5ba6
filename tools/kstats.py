"""Per-step view of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> <steps> [rows]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 40
print("kernel ms/step %.2f, launches/step %.0f" % (sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6,
                                                   sum(int(r["Calls"]) for r in rows) / steps))
for r in rows[:limit]:
    n = re.sub(r"\(anonymous namespace\)::|void ", "", r["Name"])[:100]
    print(f"{n:100s} {int(r['Calls']) / steps:7.1f} {float(r['TotalDurationNs']) / steps / 1e3:8.1f}us {float(r['AverageNs']) / 1e3:7.1f}")
