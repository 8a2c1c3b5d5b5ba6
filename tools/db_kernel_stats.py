"""Kernel statistics of a rocprofv3 run kept as a database (its default output): python tools/db_kernel_stats.py
<results.db> <out.csv> [kernel launched once per step].  Writes one row per kernel -- calls, total and average
duration, share of the kernel time, registers and LDS of the launch -- in the column layout of rocprofv3's
kernel_stats.csv, and prints the per-step view when the once-per-step kernel is named."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
once = sys.argv[3] if len(sys.argv) > 3 else None
con = sqlite3.connect(db)
rows = con.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
    "max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(workgroup_x), avg(grid_x) "
    "from kernels group by name order by sum(duration) desc").fetchall()
total = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPRs", "AccumVGPRs",
                "SGPRs", "LDSBytes", "ScratchBytes", "WorkgroupX", "AvgGridX"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], "%.1f" % r[3], "%.3f" % (100.0 * r[2] / total), r[4], r[5], r[6], r[7], r[8], r[9],
                    r[10], r[11], "%.0f" % r[12]])
if once:
    steps = sum(r[1] for r in rows if once in r[0])
    print("steps in the trace: %d; kernel time %.3f ms/step; launches %.1f/step"
          % (steps, total / steps / 1e6, sum(r[1] for r in rows) / steps))
