#!/bin/bash
# rocprofv3 passes of the HBM-bound kernels (through gpurun from the repo root): tools/profile_stream.sh r03 4194304 [LIB]
set -e
tag=${1:-r03}
P=${2:-4194304}
root=$(pwd)
out=$root/gpurun_out/${tag}_stream
mkdir -p "$out"
export TMPDIR=/tmp
if [ -n "$3" ]; then export PIR_HIP_LIB=$root/$3; fi
cd /tmp
rm -rf "$out/stats" "$out/fetch" "$out/write" "$out/sq"
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 "$root/tools/kprof_stream.py" $P 4 > "$out/stats.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/tools/kprof_stream.py" $P 2 > "$out/fetch.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 "$root/tools/kprof_stream.py" $P 2 > "$out/write.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -- python3 "$root/tools/kprof_stream.py" $P 2 > "$out/sq.log" 2>&1
python3 "$root/tools/stream_report.py" "$out/stats" "$out/fetch" "$out/write" $P "$root/gpurun_out/${tag}_stream_kernels.json" "$out/sq"
find "$out" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$root/gpurun_out/${tag}_stream_kernel_stats.csv"
rm -rf "$out/stats" "$out/fetch" "$out/write" "$out/sq"
