"""Busy / idle timeline of the last training step in a rocprofv3 kernel trace (tuning aid).

usage: timeline.py <trace dir> [window_us]"""
import csv
import glob
import re
import sys
from collections import Counter

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
win = float(sys.argv[2]) if len(sys.argv) > 2 else 500.0
# a step starts at the first edge-filter kernel of the generator's forward
starts = [s for s, e, n in rows if "filter_count" in n]
step_starts = [starts[0]]
for s in starts[1:]:
    if s - step_starts[-1] > 5e6:
        step_starts.append(s)
t0 = step_starts[-2]
t1 = step_starts[-1]
sel = [(s, e, n) for s, e, n in rows if t0 <= s < t1]
print(f"step window {(t1 - t0) / 1e3:.0f} us, {len(sel)} launches")


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |hivt::|at::native::", "", n)
    return re.sub(r"\(.*", "", n)[:48]


# union busy time and gaps
busy, cur_e, gaps = 0, t0, []
for s, e, n in sel:
    if s > cur_e:
        gaps.append((s - cur_e, cur_e, n))
        busy += e - s
        cur_e = e
    elif e > cur_e:
        busy += e - cur_e
        cur_e = e
print(f"busy (union) {busy / 1e3:.0f} us, idle {(t1 - t0 - busy) / 1e3:.0f} us, kernel time sum {sum(e - s for s, e, n in sel) / 1e3:.0f} us")
big = sorted(gaps, reverse=True)[:15]
print("largest gaps (us, at us, next kernel):")
for g, at, n in big:
    print(f"  {g / 1e3:7.1f} at {(at - t0) / 1e3:8.1f}  {short(n)}")
hist = Counter()
for g, at, n in gaps:
    hist[min(int(g / 1e3 / 2) * 2, 20)] += g / 1e3
print("idle time by gap length (us bucket -> total us):", dict(sorted(hist.items())))
nwin = int((t1 - t0) / 1e3 / win) + 1
for w in range(nwin):
    a, b = t0 + w * win * 1e3, t0 + (w + 1) * win * 1e3
    inside = [(max(s, a), min(e, b), n) for s, e, n in sel if e > a and s < b]
    tot = sum(e - s for s, e, n in inside)
    # union
    u, ce = 0, a
    for s, e, n in inside:
        if s > ce:
            u += e - s
            ce = e
        elif e > ce:
            u += e - ce
            ce = e
    names = Counter()
    for s, e, n in inside:
        names[short(n)] += e - s
    top = ", ".join(f"{k} {v / 1e3:.0f}" for k, v in names.most_common(3))
    print(f"{w * win / 1e3:6.1f} ms  busy {u / (win * 10):5.1f}%  launches {len(inside):4d}  {top}")
