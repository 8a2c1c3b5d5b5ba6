#!/bin/bash
# The bench step with single launcher variables set (alternating, to separate their effect from drift).
set -e
ms() { grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])"; }
ARGS="--gpus 1 --no-cpu-baseline --no-variant --no-tune"
for i in 1 2 3 4; do
  echo "plain $(timeout -k 10 120 python bench.py $ARGS 2>/dev/null | ms)"
  echo "RANK $(RANK=0 timeout -k 10 120 python bench.py $ARGS 2>/dev/null | ms)"
  echo "LOCAL_RANK $(LOCAL_RANK=0 timeout -k 10 120 python bench.py $ARGS 2>/dev/null | ms)"
  echo "WORLD_SIZE $(WORLD_SIZE=1 timeout -k 10 120 python bench.py $ARGS 2>/dev/null | ms)"
  echo "MASTER $(MASTER_ADDR=127.0.0.1 MASTER_PORT=29552 timeout -k 10 120 python bench.py $ARGS 2>/dev/null | ms)"
done
