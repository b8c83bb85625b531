#!/bin/bash
# experiment helper (RGB / filtered exact-order route): like tools/try_libs.sh, but the filtered-route parity tests and
# tools/rgb_breakdown.py at C5 and at C2's shape for every variant under tools/exp_libs/.
LIB=stereo-depth_amd/libstereo_mi355x.so
cp $LIB /tmp/lib_default.so
libs="$@"
[ -z "$libs" ] && libs="/tmp/lib_default.so $(ls tools/exp_libs/*.so)"
for l in $libs; do
  [ "$l" != "/tmp/lib_default.so" ] && cp "$l" $LIB
  echo "=== $l"
  timeout -k 10 300 python -m pytest tests/test_gpu_engine_rules.py -x -q -m gpu -k "filtered" 2>&1 | tail -1
  timeout -k 10 200 python tools/rgb_breakdown.py 375 1242 192 2 32 2>&1 | tail -2
  timeout -k 10 200 python tools/rgb_breakdown.py 375 1242 128 2 32 2>&1 | tail -2
  timeout -k 10 200 python tools/rgb_breakdown.py 375 1242 192 2 32 slanted 2>&1 | tail -2
  cp /tmp/lib_default.so $LIB
done
