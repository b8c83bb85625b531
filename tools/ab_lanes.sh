#!/bin/bash
# A/B of the submit modes on one box: tools/ab_lanes.sh <outdir>
O=${1:-gpurun_out/ab}; mkdir -p $O
run() { # name, env..., -- args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
show() { python3 -c "
import json,sys
r=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(r['value']), round(r['ms_per_step'],4), r['kernel_ms'])" $1; }
for rep in 1 2; do
  SMX_OVERLAP_MIN_PAIRS=0 timeout -k 10 200 python3 bench.py --quick --submit stream > $O/serial$rep.json 2>$O/err.log && show $O/serial$rep.json
  timeout -k 10 200 python3 bench.py --quick --submit stream > $O/attached$rep.json 2>$O/err.log && show $O/attached$rep.json
  timeout -k 10 200 python3 bench.py --quick --submit engine > $O/engine$rep.json 2>$O/err.log && show $O/engine$rep.json
  SMX_OVERLAP_MIN_PAIRS=0 timeout -k 10 200 python3 bench.py --quick --submit stream --engines 2 > $O/two_engines$rep.json 2>$O/err.log && show $O/two_engines$rep.json
  SMX_OVERLAP_MIN_PAIRS=0 timeout -k 10 200 python3 bench.py --quick --submit engine > $O/serial_engine$rep.json 2>$O/err.log && show $O/serial_engine$rep.json
done
