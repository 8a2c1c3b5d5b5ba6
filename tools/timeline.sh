#!/bin/bash
# Kernel trace of a short bench run reduced to a busy / idle timeline of the last step.
set -e
R=$(pwd)
OUT=$R/gpurun_out/timeline
SCR=/tmp/timeline_run
mkdir -p "$OUT" "$SCR"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$SCR" -o t -- python3 "$R/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-tune --no-variant > "$OUT/bench.log" 2>&1
python3 "$R/tools/timeline.py" "$SCR" > "$OUT/timeline.txt"
echo done
