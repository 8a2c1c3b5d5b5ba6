#!/bin/bash
# Kernel-trace statistics, per-replay statistics, the matrix-core utilisation pass and the two HBM-traffic counter
# passes of the default bench workload (one rocprofv3 run per pass, the program directly after `--`).
# Run on the GPU box from the repository root:  bash tools/profile_bench.sh <tag>
set -e
TAG=${1:-run}
R=$(pwd)
OUT=$R/gpurun_out/prof_$TAG
SCR=/tmp/prof_$TAG
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$SCR/stats" -o stats -- python3 "$R/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-tune --no-variant --no-pool-fitted > "$OUT/bench_stats.log" 2>&1
cp $(find "$SCR/stats" -name "*kernel_stats.csv") "$OUT/kernel_stats.csv"
echo "stats pass done"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$SCR/replay" -o t -- python3 "$R/bench.py" --steps 20 --no-cpu-baseline --no-tune --no-variant --no-pool-fitted --no-kernel-timing > "$OUT/bench_replay.log" 2>&1
python3 "$R/tools/replay_stats.py" "$SCR/replay" 20 400 > "$OUT/replay_kernel_stats.txt"
echo "replay pass done"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$SCR/mfma" -o mfma -- python3 "$R/bench.py" --steps 2 --warmup 1 --pool 4 --no-cpu-baseline --no-tune --no-variant --no-pool-fitted --no-kernel-timing > "$OUT/bench_mfma.log" 2>&1
python3 "$R/tools/summarize_pmc.py" "$SCR/mfma" > "$OUT/pmc_mfma.csv"
echo "mfma pass done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$SCR/fetch" -o fetch -- python3 "$R/bench.py" --steps 2 --warmup 1 --pool 4 --no-cpu-baseline --no-tune --no-variant --no-pool-fitted > "$OUT/bench_fetch.log" 2>&1
python3 "$R/tools/summarize_pmc.py" "$SCR/fetch" > "$OUT/pmc_fetch_size.csv"
echo "fetch pass done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$SCR/write" -o write -- python3 "$R/bench.py" --steps 2 --warmup 1 --pool 4 --no-cpu-baseline --no-tune --no-variant --no-pool-fitted > "$OUT/bench_write.log" 2>&1
python3 "$R/tools/summarize_pmc.py" "$SCR/write" > "$OUT/pmc_write_size.csv"
echo "write pass done"
ls -la "$OUT"
