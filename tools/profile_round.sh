#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root):  tools/profile_round.sh r02
#   1. rocprofv3 --kernel-trace --stats of the bench (4 steps + 1 warm-up, no CPU baseline)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/kprof.py (forward + get_all) and tools/kprof_bwd.py
# Outputs under gpurun_out/<tag>_prof/; copy the summaries into profiles/ afterwards.
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/${tag}_prof
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/bench" -- python3 "$root/bench.py" --steps 4 --warmup 1 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/bench_under_rocprof.err"
echo "bench traced"
for prog in kprof kprof_bwd; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d "$out/${prog}_$ctr" -- python3 "$root/tools/$prog.py" > "$out/${prog}_$ctr.log" 2>&1
    echo "$prog $ctr done"
  done
done
find "$out" -name "*kernel_stats.csv" -o -name "*counter_collection.csv" | head -20
