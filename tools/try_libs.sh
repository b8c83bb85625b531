#!/bin/bash
# experiment helper: for every prebuilt variant of the library under tools/exp_libs/ (built in the container with
# SMX_* tuning macros; .so files are git-ignored) run one full-size C2 parity test and the serial + pipelined bench regions.
# Usage (on the GPU box): bash tools/try_libs.sh [variant.so ...]   -- restores the default library afterwards.
LIB=stereo-depth_amd/libstereo_mi355x.so
cp $LIB /tmp/lib_default.so
libs="$@"
[ -z "$libs" ] && libs="/tmp/lib_default.so $(ls tools/exp_libs/*.so)"
for l in $libs; do
  [ "$l" != "/tmp/lib_default.so" ] && cp "$l" $LIB
  echo "=== $l"
  [ -z "$SKIP_PARITY" ] && timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "C2 or benchmarked or worst_case" 2>&1 | tail -1
  timeout -k 10 200 python bench.py --steps 30 --warmup 3 --quick --repeats 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/s %.0f  serial %.0f  ms/step %.3f  kernels %s' % (d['value'], d.get('value_serial') or 0, d['ms_per_step'], d['kernel_ms']))"
  cp /tmp/lib_default.so $LIB
done
