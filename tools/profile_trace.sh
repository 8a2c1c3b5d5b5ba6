#!/bin/bash
# Per-dispatch kernel trace of a short bench run, reduced to the hand-written linear kernels by grid.
set -e
R=$(pwd)
OUT=$R/gpurun_out/trace
SCR=/tmp/trace_run
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$SCR" -o t -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-tune --no-variant > "$OUT/bench.log" 2>&1
python3 - "$SCR" > "$OUT/linear_by_grid.txt" <<'PY'
import csv, glob, sys, re
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        if "hivt" not in n:
            continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
        key = (m.group(1) + (m.group(2) or ""), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
        acc[key][0] += 1
        acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
steps = 6
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:70]:
    print(f"{k[0]:60s} grid {k[1]:>8s} {k[2]:>3s} {k[3]:>4s}  calls/step {n / steps:6.1f}  us/step {t / steps:8.1f}  avg {t / n:7.1f}")
PY
echo done
