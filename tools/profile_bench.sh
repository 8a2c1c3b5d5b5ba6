#!/bin/bash
# Kernel-trace statistics and the two HBM-traffic counter passes of the default bench workload.
# Run on the GPU box from the repository root:  bash tools/profile_bench.sh <tag>
set -e
TAG=${1:-run}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/stats" -o stats -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-tune --no-variant > "$OUT/bench_stats.log" 2>&1
cp $(find "$SCR/stats" -name "*kernel_stats.csv") "$OUT/kernel_stats.csv"
echo "stats pass done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$SCR/fetch" -o fetch -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-tune --no-variant > "$OUT/bench_fetch.log" 2>&1
python3 "$R/tools/summarize_pmc.py" "$SCR/fetch" > "$OUT/pmc_fetch_size.csv"
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$SCR/write" -o write -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-tune --no-variant > "$OUT/bench_write.log" 2>&1
python3 "$R/tools/summarize_pmc.py" "$SCR/write" > "$OUT/pmc_write_size.csv"
echo "write pass done"
ls -la "$OUT"
