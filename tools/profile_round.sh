#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel trace + the two PMC passes of the bench's
# roofline region (`--quick --repeats 1 --submit stream`: the caller's stream, 64 pairs per launch), each in its own
# run (gpurun refuses --pmc combined with other trace domains), plus a kernel trace of the pipelined
# timed region (two 32-pair launches per kernel and step on two streams).
#   tools/profile_round.sh r01
TAG=${1:-r01}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O; cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 10 --warmup 2 --quick --repeats 1 --submit stream > $O/trace_bench.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --quick --repeats 1 --submit stream > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 3 --warmup 1 --quick --repeats 1 --submit stream > $O/pmc_write.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pipelined -- python3 bench.py --steps 10 --warmup 2 --quick --repeats 1 --submit engine --no-serial-pass > $O/trace_pipelined.log 2>&1 || exit 1
cp $(ls $O/trace_pipelined/*/*_kernel_stats.csv | head -1) $O/summary_pipelined_kernel_stats.csv 2>/dev/null
grep "^{\"metric\"" $O/trace_pipelined.log | tail -1 > $O/bench_pipelined_under_rocprof.json
grep "^{\"metric\"" $O/trace_bench.log | tail -1 > $O/bench_under_rocprof.json
python3 tools/pmc_traffic.py $O $O/summary $TAG 64
