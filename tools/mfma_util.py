"""Matrix-core utilisation per kernel from one counter pass (tools/summarize_pmc.py output with
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE): python tools/mfma_util.py <pmc_mfma.csv>.
SQ_VALU_MFMA_BUSY_CYCLES is the sum over the chip's 1 024 SIMDs of the cycles their matrix unit was busy;
GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the dispatch was active.  utilisation = busy cycles /
(active cycles / 8 * 1 024)."""
import csv
import re
import sys
from collections import defaultdict

by = defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    by[r["kernel"]][r["counter"]] = (float(r["mean_per_dispatch"]), int(r["dispatches"]))
print(f"{'kernel':84s} {'dispatches':>10s} {'active us @2.4GHz':>18s} {'mfma busy / SIMD-cycles':>24s}")
rows = []
for k, v in by.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in v or "GRBM_GUI_ACTIVE" not in v:
        continue
    busy, n = v["SQ_VALU_MFMA_BUSY_CYCLES"]
    active = v["GRBM_GUI_ACTIVE"][0] / 8.0
    if busy <= 0:
        continue
    rows.append((busy / (active * 1024.0), re.sub(r"\(anonymous namespace\)::|void |hivt::", "", k)[:84], n, active / 2400.0))
for util, name, n, us in sorted(rows, reverse=True):
    print(f"{name:84s} {n:10d} {us:18.1f} {util:24.3f}")
