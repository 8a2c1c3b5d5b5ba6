#!/bin/bash
# The per-replay kernel statistics alone (the A/B instrument between two trees): one kernel-trace run of the default
# bench workload, cut into steps by tools/replay_stats.py.   bash tools/replay_pass.sh <tag>
set -e
TAG=${1:-run}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$SCR/replay" -o t -- python3 "$R/bench.py" --steps 20 --no-cpu-baseline --no-tune --no-variant --no-pool-fitted --no-kernel-timing > "$OUT/bench_replay.log" 2>&1
python3 "$R/tools/replay_stats.py" "$SCR/replay" 20 400 > "$OUT/replay_kernel_stats.txt"
head -5 "$OUT/replay_kernel_stats.txt"
