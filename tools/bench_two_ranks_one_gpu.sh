#!/bin/bash
# bench.py with two data-parallel ranks that share cuda:0 over gloo: the multi-rank control flow of the bench
# (capacity agreement, recorded segments around the all-reduces, max-over-ranks timing) without a second GPU.
# The number it prints is not a throughput of anything; rank 0's JSON line goes to gpurun_out/bench_2ranks.json.
set -e
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 WORLD_SIZE=2 LOCAL_RANK=0 HIVT_DIST_BACKEND=gloo
ARGS="--gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-variant --no-tune --scenes 32"
RANK=1 timeout -k 10 240 python bench.py $ARGS > gpurun_out/bench_2ranks_r1.log 2>&1 &
P1=$!
RANK=0 timeout -k 10 240 python bench.py $ARGS > gpurun_out/bench_2ranks.json 2> gpurun_out/bench_2ranks_r0.err
R0=$?
wait $P1
R1=$?
echo "rank 0 exit $R0, rank 1 exit $R1"
tail -c 400 gpurun_out/bench_2ranks.json
exit $((R0 + R1))
