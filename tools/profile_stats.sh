#!/bin/bash
# Kernel-trace statistics of the default bench workload: bash tools/profile_stats.sh <tag>
set -e
TAG=${1:-run}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/stats" -o stats -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-tune --no-variant $BENCH_ARGS > "$OUT/bench_stats.log" 2>&1
cp $(find "$SCR/stats" -name "*kernel_stats.csv") "$OUT/kernel_stats.csv"
echo "stats pass done"
