#!/bin/bash
# A/B of library variants on ONE box: bash tools/ab_libs.sh "<bench args>" variant [variant ...]
# ("default" = the product library; others: python stereo-depth_amd/build.py --variant=NAME with SMX_EXTRA_FLAGS).
# Two alternating rounds; prints the fields of the bench line that matter for kernel work.
ARGS="$1"; shift
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = "default" ]; then unset SMX_LIB_PATH; else export SMX_LIB_PATH=$PWD/stereo-depth_amd/libstereo_mi355x.$v.so; fi
    timeout -k 10 300 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
keys = ['value', 'value_serial', 'value_noise', 'value_slanted', 'value_real', 'value_real_rgb', 'value_rgb', 'single_pair_latency_us']
print('$v round $round:', ' '.join('%s=%.0f' % (k.replace('value_', '').replace('single_pair_latency_us', 'lat_us'), d[k]) for k in keys if d.get(k)), d.get('kernel_ms'))
c = d.get('configs')
if c: print('   configs:', ' | '.join('%s %.0f/%.0fus/%.0f' % (k.split()[0] + ('r' if 'RGB' in k else ''), v['pairs_per_s'], v['single_call_latency_us'], v.get('single_calls_pipelined_per_s', 0)) for k, v in c.items()))"
  done
done
