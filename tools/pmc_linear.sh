#!/bin/bash
set -e
R=$(pwd)
OUT=$R/gpurun_out/pmc_linear
SCR=/tmp/pmc_linear
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d "$SCR/p1" -o p1 -- python3 "$R/tools/linear_probe.py" > "$OUT/p1.log" 2>&1
python3 - "$SCR/p1" > "$OUT/pmc.txt" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "hivt" in row["Kernel_Name"] and "linear" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:70], row["Grid_Size"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: sum(v) / len(v) for n, v in c.items()})
PY
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/p2" -o p2 -- python3 "$R/tools/linear_probe.py" > "$OUT/p2.log" 2>&1
cp $(find "$SCR/p2" -name "*kernel_trace.csv") "$OUT/trace.csv"
echo done
