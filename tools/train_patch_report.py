#!/usr/bin/env python3
"""GPU-busy fraction and launches/step of tools/bench_train_patch.py from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/bench_train_patch.py ... > run.json
    python tools/train_patch_report.py DIR run.json [--top 12]

The timed region of the traced program is bracketed by two `erfinv` marker kernels.  Between them:
  wall          = start of the closing marker - end of the opening marker   (device clock)
  kernel_time   = sum of the dispatch durations
  busy_union    = length of the union of the dispatch intervals (overlapping kernels counted once)
  launches/step = dispatches / steps
Prints one JSON line (the run's own line + these figures + the top kernels by time and by launches).
"""
import csv
import glob
import json
import os
import sys


def main():
    d, run = sys.argv[1], sys.argv[2]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    line = [ln for ln in open(run) if ln.startswith("{")][-1]
    info = json.loads(line)
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "erfinv" in r[2]]
    if len(marks) < 2:
        raise SystemExit(f"expected two erfinv markers in the trace, found {len(marks)}")
    a, b = marks[0], marks[-1]
    region = rows[a + 1:b]
    wall = rows[b][0] - rows[a][1]
    ktime = sum(e - s for s, e, _ in region)
    union, cur_s, cur_e = 0, None, None
    for s, e, _ in region:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        union += cur_e - cur_s
    gaps = []
    prev_e = rows[a][1]
    for s, e, _ in region:
        if s > prev_e:
            gaps.append(s - prev_e)
        prev_e = max(prev_e, e)
    gaps.sort(reverse=True)
    by = {}
    for s, e, n in region:
        n = n.split("(")[0][:80]
        t = by.setdefault(n, [0, 0])
        t[0] += e - s
        t[1] += 1
    steps = info["steps"]
    info.update({
        "wall_ms_per_step_device_clock": round(wall / steps / 1e6, 3),
        "kernel_ms_per_step": round(ktime / steps / 1e6, 3),
        "gpu_busy_fraction": round(union / wall, 4),
        "launches_per_step": round(len(region) / steps, 1),
        "idle_ms_per_step": round((wall - union) / steps / 1e6, 3),
        "idle_gaps_over_100us_per_step": round(sum(1 for g in gaps if g > 100e3) / steps, 1),
        "idle_ms_in_gaps_over_100us_per_step": round(sum(g for g in gaps if g > 100e3) / steps / 1e6, 3),
        "top_by_time": [{"kernel": k, "ms_per_step": round(v[0] / steps / 1e6, 3), "launches_per_step": round(v[1] / steps, 1)}
                        for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:top]],
        "top_by_launches": [{"kernel": k, "launches_per_step": round(v[1] / steps, 1), "ms_per_step": round(v[0] / steps / 1e6, 3)}
                            for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:top]],
    })
    print(json.dumps(info))


if __name__ == "__main__":
    main()
