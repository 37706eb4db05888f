#!/bin/bash
# Everything the round's profiles/ entries come from, in one gpurun call (from the repo root): tools/profile_round_all.sh r03
#   1. tools/profile_round.sh  : bench under rocprofv3 --kernel-trace --stats + FETCH_SIZE / WRITE_SIZE passes
#   2. tools/pmc_traffic.py    : HBM bytes per point per kernel -> profiles/<tag>_pmc_traffic.json
#   3. tools/profile_mfma.sh   : matrix-pipe busy share and effective clock -> profiles/<tag>_mfma_util.json
set -e
tag=${1:-r03}
root=$(pwd)
tools/profile_round.sh $tag
p=$root/gpurun_out/${tag}_prof
f() { find "$1" -name "*counter_collection.csv" | head -1; }
PMC_OUT=${tag}_pmc_traffic.json python3 tools/pmc_traffic.py "$(f $p/kprof_FETCH_SIZE)" "$(f $p/kprof_WRITE_SIZE)" 2097152
PMC_OUT=${tag}_pmc_traffic.json python3 tools/pmc_traffic.py "$(f $p/kprof_bwd_FETCH_SIZE)" "$(f $p/kprof_bwd_WRITE_SIZE)" 524288
cp "$(find $p/bench -name '*kernel_stats.csv' | head -1)" gpurun_out/${tag}_bench_kernel_stats.csv
cp $p/bench_under_rocprof.json gpurun_out/${tag}_bench_under_rocprof.json
cp profiles/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc_traffic.json
tools/profile_mfma.sh $tag
python3 tools/mfma_util.py "$(f $root/gpurun_out/${tag}_mfma/pass)" gpurun_out/${tag}_mfma_util.json
rm -rf $p/bench $root/gpurun_out/${tag}_mfma/pass
