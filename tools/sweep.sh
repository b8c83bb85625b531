#!/bin/bash
# experiment helper: rebuild the library with different compile-time knobs and bench each
for flags in "$@"; do
  SMX_EXTRA_FLAGS="$flags" python stereo-depth_amd/build.py --force > /dev/null || exit 1
  echo "=== $flags"
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --latency 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/s %.0f  ms/step %.3f  kernels %s  lat_us %.0f' % (d['value'], d['ms_per_step'], d['kernel_ms'], d['single_pair_latency_us']))"
done
