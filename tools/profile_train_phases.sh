#!/bin/bash
# Per-phase table of the training step (run through gpurun from the repo root): tools/profile_train_phases.sh r03 "64 128"
set -e
tag=${1:-r03}
patches=${2:-"64 128"}
root=$(pwd)
out=$root/gpurun_out/${tag}_train_phases
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for p in $patches; do
  rm -rf "$out/trace_p$p"
  timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_p$p" -- python3 "$root/tools/train_patch_phases.py" --patch $p --labels "$out/labels_p$p.json" > "$out/run_p$p.log" 2>&1
  python3 "$root/tools/train_patch_phases.py" --report "$out/trace_p$p" "$out/labels_p$p.json" >> "$root/gpurun_out/${tag}_train_phases.jsonl"
  rm -rf "$out/trace_p$p"
  echo "phases patch $p done"
done
