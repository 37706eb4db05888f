#!/usr/bin/env python3
"""Per-phase launch / kernel-time / wall table of the training step (diagnosis for VERDICT r2 item 2).

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/train_patch_phases.py --patch 128 --labels L.json
    python tools/train_patch_phases.py --report DIR L.json

The traced program runs tools/bench_train_patch.py's state and emits one `erfinv` marker kernel at the start of every
phase of every timed step (trace + edge walk, visibility, direct shading, indirect bounce, edge pixels, losses,
backward, optimizers); the i-th marker in the trace carries the i-th label of L.json.  --report sums, per phase,
the dispatches, their durations and the device-clock wall between markers.
"""
import argparse
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def report(d, labels_path):
    labels = json.load(open(labels_path))
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "erfinv" in r[2]]
    assert len(marks) == len(labels["labels"]), (len(marks), len(labels["labels"]))
    steps = labels["steps"]
    acc = {}
    for j in range(len(marks) - 1):
        lab = labels["labels"][j]
        if lab == "end":
            continue
        a, b = marks[j], marks[j + 1]
        t = acc.setdefault(lab, {"launches": 0, "kernel_ns": 0, "wall_ns": 0, "pir_ns": 0, "pir_launches": 0})
        t["wall_ns"] += rows[b][0] - rows[a][1]
        for s, e, n in rows[a + 1:b]:
            t["launches"] += 1
            t["kernel_ns"] += e - s
            if "pir::" in n:
                t["pir_ns"] += e - s
                t["pir_launches"] += 1
    out = {"patch": labels["patch"], "stage": labels["stage"], "steps": steps, "phases": {}}
    for lab, t in acc.items():
        out["phases"][lab] = {"launches_per_step": round(t["launches"] / steps, 1),
                              "hip_kernel_launches_per_step": round(t["pir_launches"] / steps, 1),
                              "kernel_ms_per_step": round(t["kernel_ns"] / steps / 1e6, 3),
                              "hip_kernel_ms_per_step": round(t["pir_ns"] / steps / 1e6, 3),
                              "wall_ms_per_step": round(t["wall_ns"] / steps / 1e6, 3),
                              "idle_ms_per_step": round((t["wall_ns"] - t["kernel_ns"]) / steps / 1e6, 3)}
    print(json.dumps(out))


def main():
    if "--report" in sys.argv:
        i = sys.argv.index("--report")
        return report(sys.argv[i + 1], sys.argv[i + 2])
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--stage", type=int, default=2)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--labels", required=True)
    args = ap.parse_args()
    import numpy as np
    import torch
    import bench_train_patch as B
    from pir_amd import train
    from pir_amd.renderer import raytracer as rt
    state = B.build_state("cuda", args.patch, args.stage)
    corners = B.object_crops(800, args.patch, args.warmup + args.steps, seed=3)
    it = {"i": 0, "on": False}
    labels = []
    marker = torch.full((1,), 0.5, device="cuda")

    def mark(label):
        if it["on"]:
            torch.erfinv(marker)
            labels.append(label)

    orig_crop = rt.Camera.crop_region
    rt.Camera.crop_region = lambda self, trgt_W, trgt_H, center_crop=False, ul_corner=None, images=None: \
        orig_crop(self, trgt_W, trgt_H, ul_corner=corners[it["i"] % len(corners)], images=images)

    depth = [0]

    def wrap(mod, name, label):
        f = getattr(mod, name)

        def g(*a, **k):
            if depth[0] == 0:        # nested calls (render_edge_pixels shades both sides itself) keep the outer label
                mark(label)
            depth[0] += 1
            try:
                return f(*a, **k)
            finally:
                depth[0] -= 1
        setattr(mod, name, g)
    wrap(rt, "raytrace_camera", "trace + edge walk")
    wrap(rt, "render_light_visibility", "visibility march (forward)")
    wrap(rt, "render_normal_and_color", "direct shading (forward)")
    wrap(rt, "render_indirect_lighting", "indirect bounce (forward)")
    wrap(rt, "render_edge_pixels", "edge pixels (forward)")

    sdf_grad = state.sdf_network.gradient

    def gradient(x):
        mark("eikonal samples + losses")
        return sdf_grad(x)
    state.sdf_network.gradient = gradient
    orig_backward = torch.Tensor.backward

    def backward(self, *a, **k):
        mark("backward")
        r = orig_backward(self, *a, **k)
        mark("optimizers + status read-back")
        return r
    torch.Tensor.backward = backward

    np.random.seed(4)
    torch.manual_seed(5)
    for i in range(args.warmup):
        it["i"] = i
        train.training_step(state, 12000)
    torch.cuda.synchronize()
    it["on"] = True
    for i in range(args.steps):
        it["i"] = args.warmup + i
        mark("zero_grad + crop")
        train.training_step(state, 12000)
        mark("end")
    torch.cuda.synchronize()
    json.dump({"labels": labels, "steps": args.steps, "patch": args.patch, "stage": args.stage}, open(args.labels, "w"))
    print("labels written:", len(labels))


if __name__ == "__main__":
    main()
