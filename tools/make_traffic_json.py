"""Turn the per-kernel means of the FETCH_SIZE / WRITE_SIZE passes (tools/summarize_pmc.py output)
into profiles/<tag>_seg_attn_traffic.json: HBM bytes per launch of every segment-attention kernel.

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests
of wide streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane
stores."""
import csv
import json
import re
import sys

src, tag, embed_dim, scenes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
def caller(rows, split, wpn, unroll):
    """Which module launches a kernel instance on the benchmark's workload.  `rows` is the kernel's row-source
    template argument: 1 = node terms gathered in the kernel (the interaction module before its node rows moved
    into the projection's epilogue), 2 = interleaved rows -> the temporal encoder's last layer; plain rows: one
    row per lane slot -> agent-agent stage (1.3 edges per row), four waves per row with eight edges in flight ->
    lane-actor stage (52 edges per row), the remaining shapes -> the interaction module (18 edges per row)."""
    if rows in ("1", "true"):
        return "GlobalInteractorLayer"
    if rows == "2":
        return "TemporalEncoderLayer"
    if split == "true":
        return "AAEncoder"
    if wpn == "4" and unroll == "8":
        return "ALEncoder"
    return "GlobalInteractorLayer" if rows == "0" else "TemporalEncoderLayer"
out = {}
for name, scale in (("pmc_fetch_size.csv", 2 * 1024.0), ("pmc_write_size.csv", 1024.0)):
    for r in csv.DictReader(open(f"{src}/{name}")):
        m = re.search(r"seg_attn_(fwd|bwd)_kernel<(\d+), (\d+), (\d+), (\w+), (\w+)(?:, (\d+))?(?:, (?:true|false)){0,2}>", r["kernel"])
        if not m:
            continue
        kind, _d, _lph, wpn, glob, split, unroll = m.groups()
        key = f"{kind}/{caller(glob, split, wpn, unroll)}"
        e = out.setdefault(key, {"read_bytes": 0.0, "write_bytes": 0.0})
        e["read_bytes" if "fetch" in name else "write_bytes"] += float(r["mean_per_dispatch"]) * scale
for e in out.values():
    e["hbm_bytes"] = e["read_bytes"] + e["write_bytes"]
json.dump({"workload": {"embed_dim": embed_dim, "scenes_per_gpu": scenes},
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of bench.py; KiB -> bytes, "
                     "FETCH_SIZE doubled (gfx950 counts 128-byte read requests as 64 bytes)",
           "per_launch": out}, open(f"profiles/{tag}_seg_attn_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
