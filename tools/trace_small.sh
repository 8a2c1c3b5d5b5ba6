#!/bin/bash
set -e
R=$(pwd)
OUT=$R/gpurun_out/trace_small
SCR=/tmp/trace_small
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d "$SCR" -o t -- python3 "$R/tools/linear_probe_small.py" > "$OUT/log.txt" 2>&1
python3 - "$SCR" > "$OUT/small.txt" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
acc = defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "hivt" not in n: continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
        acc[(m.group(1) + (m.group(2) or ""), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print(k, "n", len(v), "min %.1f med %.1f" % (v[0], v[len(v)//2]))
PY
