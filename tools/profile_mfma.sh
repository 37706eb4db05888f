#!/bin/bash
# MFMA utilisation and effective clock of the three dominant kernels (run through gpurun from the repo root):
#   tools/profile_mfma.sh r02
# One rocprofv3 --pmc pass (6 SQ + 1 GRBM counters, within the per-block slot limits) with --kernel-trace for the
# dispatch durations; tools/mfma_util.py turns the two CSVs into profiles/<tag>_mfma_util.json.
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/${tag}_mfma
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$out/pass" -- python3 "$root/tools/kprof_mfma.py" > "$out/pass.log" 2>&1
echo "pmc pass done"
find "$out" -name "*.csv"
