#!/bin/bash
# experiment helper: A/B an environment switch of the library on the bench's timed regions.
#   bash tools/ab_env.sh SMX_SIDE_STREAM 0 1      (three alternating rounds)
VAR=$1; shift
for round in 1 2 3; do
  for v in "$@"; do
    echo -n "$VAR=$v  "
    env $VAR=$v timeout -k 10 200 python bench.py --steps 50 --warmup 5 --quick --repeats 1 --no-serial-pass 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/s %.0f  ms/step %.4f' % (d['value'], d['ms_per_step']))"
  done
done
