#!/bin/bash
# Training step at the reference's patch sizes under rocprofv3 --kernel-trace (run through gpurun from the repo root):
#   tools/profile_train_patch.sh r03 "64 128 256" "2 3"
# For each (patch, stage): wall clock untraced (40 steps after 40 warm-up steps: the caching allocator and MIOpen need
# about 30 steps of varying hit counts to reach the steady state a 50 000-step run is in), then a traced run; tools/train_patch_report.py turns the trace into
# GPU-busy fraction, launches/step and the top kernels.  One JSON line per case into gpurun_out/<tag>_train_patch.jsonl
set -e
tag=${1:-r03}
patches=${2:-"64 128 256"}
stages=${3:-"2 3"}
root=$(pwd)
out=$root/gpurun_out/${tag}_train_patch
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
for st in $stages; do
  for p in $patches; do
    timeout -k 10 240 python3 "$root/tools/bench_train_patch.py" --patch $p --stage $st --steps 40 --warmup 40 > "$out/untraced_s${st}_p${p}.json" 2> "$out/untraced_s${st}_p${p}.err"
    rm -rf "$out/trace_s${st}_p${p}"
    timeout -s KILL 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace_s${st}_p${p}" -- python3 "$root/tools/bench_train_patch.py" --patch $p --stage $st --steps 10 --warmup 30 > "$out/traced_s${st}_p${p}.json" 2> "$out/traced_s${st}_p${p}.err"
    python3 "$root/tools/train_patch_report.py" "$out/trace_s${st}_p${p}" "$out/traced_s${st}_p${p}.json" > "$out/report_s${st}_p${p}.json"
    python3 - "$out/untraced_s${st}_p${p}.json" "$out/report_s${st}_p${p}.json" >> "$root/gpurun_out/${tag}_train_patch.jsonl" <<'PY'
import json, sys
u = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = json.loads(open(sys.argv[2]).read())
r["untraced"] = {k: u[k] for k in ("it_per_s", "rays_per_s", "ms_per_step", "hits_per_step", "edge_pixels_per_step", "steps", "warmup")}
print(json.dumps(r))
PY
    rm -rf "$out/trace_s${st}_p${p}"     # the raw trace is large; the report keeps what is needed
    echo "stage $st patch $p done"
  done
done
