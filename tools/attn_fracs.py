"""Prints the per-launch durations / roofline fractions of the six segment-attention launch kinds from a
bench.py JSON line on stdin."""
import json
import sys

d = json.loads(sys.stdin.read())
print("ms/step %.2f" % d["ms_per_step"])
for e in d["edge_attention_kernels"]:
    print(f"{e['kernel']:22s} {e['caller']:24s} {e['avg_launch_us']:7.1f} us  frac {e['frac']:.3f}")
