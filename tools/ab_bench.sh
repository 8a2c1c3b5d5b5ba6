#!/bin/bash
# A/B of an environment switch on one box: bash tools/ab_bench.sh VAR=VALUE [rounds]
# prints ms/step of bench.py with and without the setting, alternating.
set -e
SETTING=$1
ROUNDS=${2:-2}
for i in $(seq $ROUNDS); do
  a=$(timeout -k 10 200 python bench.py --no-cpu-baseline --no-tune --no-variant 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(env $SETTING timeout -k 10 200 python bench.py --no-cpu-baseline --no-tune --no-variant 2>/dev/null | python -c "import json,sys; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
  echo "default $a ms/step | $SETTING $b ms/step"
done
