"""Per-step kernel statistics of the LAST steps in a rocprofv3 kernel trace (csv): python tools/replay_stats.py
<trace dir> [steps=10] [rows=60].  A step starts at its batch_prepare_kernel launch; the last `steps` complete
steps are aggregated: launches and microseconds per step of every kernel, busy and idle time of the step.
Run the bench with --no-kernel-timing so that the last steps of the trace are the timed graph replays."""
import csv
import glob
import re
import sys
from collections import defaultdict

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 60
starts = [i for i, (_, _, n) in enumerate(rows) if "batch_prepare_kernel" in n]
lo, hi = starts[-steps - 1], starts[-1]
sel = rows[lo:hi]
busy = sum(e - s for s, e, _ in sel)
span = rows[hi][0] - rows[lo][0]
# overlap-free busy time (kernels on one stream do not overlap; kept general)
end, covered = 0, 0
for s, e, _ in sel:
    if e > end:
        covered += e - max(s, end)
        end = e
print("last %d steps: %.3f ms/step wall, %.3f ms/step kernel time (%.3f covered), %.1f launches/step, idle %.3f ms/step"
      % (steps, span / steps / 1e6, busy / steps / 1e6, covered / steps / 1e6, len(sel) / steps,
         (span - covered) / steps / 1e6))
agg = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    a = agg[n]
    a[0] += 1
    a[1] += e - s
fam = defaultdict(float)
for n, (c, t) in agg.items():
    key = ("framework" if ("at::native" in n or "rocclr" in n) else
           "dense" if re.search(r"linear_|critic_chain|embed2|reduce_group", n) else
           "edge attention" if "seg_attn" in n else
           "temporal attention" if "temporal_attn" in n else
           "norms / dropout / gates" if re.search(r"ln_|dropout|gated|reduce_rows", n) else "other")
    fam[key] += t
print("families (us/step):", {k: round(v / steps / 1e3, 1) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:limit]:
    name = re.sub(r"\(anonymous namespace\)::|void |hivt::", "", n)[:96]
    print(f"{name:96s} {c / steps:7.1f} {t / steps / 1e3:8.1f}us {t / c / 1e3:7.1f}")
# every launch of the last step, in launch order, for kernels named on the command line after the numbers:
# python tools/replay_stats.py <dir> 20 60 ln_bwd linear_fwd
if len(sys.argv) > 4:
    last = rows[starts[-2]:starts[-1]]
    for needle in sys.argv[4:]:
        print("launches of the last step matching %r (position in the step, microseconds):" % needle)
        print(" ".join("%d:%.1f" % (i, (e - s) / 1e3) for i, (s, e, n) in enumerate(last) if needle in n))
