"""Per (kernel, grid size) view of a rocprofv3 kernel trace: launches, average and total duration -- which shapes of a
templated kernel the time goes to.

    python tools/trace_by_grid.py <dir with *_kernel_trace.csv> [rows]
"""
import collections
import csv
import glob
import sys

src = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
path = sorted(glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True))[0]
tot = collections.Counter()
cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    key = (r["Kernel_Name"][:110], int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1) * int(r.get("Grid_Size_Z", 1) or 1)
           // max(int(r["Workgroup_Size_X"]), 1))
    tot[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[key] += 1
for key, us in tot.most_common(top):
    print(f"{us:10.0f} us  {cnt[key]:5d} x {us / cnt[key]:7.1f} us  blocks {key[1]:7d}  {key[0]}")
