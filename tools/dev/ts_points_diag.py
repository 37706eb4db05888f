#!/usr/bin/env python3
"""Diagnosis: where does the training-step fixture differ on the GPU?  Runs the step (a) as is, (b) with the primary
hit points pinned to the fp32 reference's, (c) with eager material MLPs, and prints the point deviations and the
worst material-gradient errors against the fp32 / fp64 reference."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import train_step_util  # noqa: E402
from conftest import load_golden  # noqa: E402
from full800_util import elem_err  # noqa: E402


def run(mode):
    from pir_amd import train, mlp_hip
    from pir_amd.renderer import raytracer as rt
    d, state = train_step_util.build_state("cuda")
    c = load_golden("composite_fp64.npz")
    orig = rt.raytrace_camera
    if "pin" in mode:
        def pinned(*a, **k):
            res = orig(*a, **k)
            m = torch.from_numpy(d["out.mask"]).cuda() & res["convergent_mask"]
            ref_p = torch.from_numpy(d["out.points"]).cuda()
            res["points"] = torch.where(m[..., None], ref_p, res["points"])
            return res
        rt.raytrace_camera = pinned
    np.random.seed(int(d["np_seed"]))
    eik = torch.from_numpy(d["eik_points"]).cuda()
    try:
        if "eager" in mode:
            with mlp_hip.eager_grad():
                out = train.training_step(state, int(d["global_step"]), eik_points=eik)
        else:
            out = train.training_step(state, int(d["global_step"]), eik_points=eik)
    finally:
        rt.raytrace_camera = orig
    m = d["out.mask"]
    pts = out["results"]["points"].detach().cpu().numpy()
    dev32 = np.abs(pts - d["out.points"]).max(-1)[m]
    ref_dev = np.abs(d["out.points"] - c["ts.out.points"]).max(-1)[m]
    dev64 = np.abs(pts - c["ts.out.points"]).max(-1)[m]
    q = lambda v: [float(np.quantile(v, x)) for x in (0.5, 0.9, 0.99, 1.0)]  # noqa: E731
    print(f"[{mode}] hit points (n={m.sum()}): |hip-ref32| q50/90/99/max {q(dev32)}; |hip-ref64| {q(dev64)}; "
          f"|ref32-ref64| {q(ref_dev)}")
    print(f"[{mode}] points farther than 1e-6: hip-ref32 {(dev32 > 1e-6).sum()}, hip-ref64 {(dev64 > 1e-6).sum()}, "
          f"ref32-ref64 {(ref_dev > 1e-6).sum()};  >1e-5: {(dev32 > 1e-5).sum()} / {(dev64 > 1e-5).sum()} / {(ref_dev > 1e-5).sum()}")
    col = out["results"]["color"].detach().cpu().numpy()
    print(f"[{mode}] color: vs ref32 {elem_err(col, d['out.color']):.2e} vs fp64 {elem_err(col, c['ts.out.color']):.2e} "
          f"ref32 vs fp64 {elem_err(d['out.color'], c['ts.out.color']):.2e}")
    nets = state.color_network_dict
    for nm in ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
        worst = (0, None)
        for k, p in nets[nm].named_parameters():
            ref, r64 = d[f"g.{nm}.{k}"], c[f"ts.g.{nm}.{k}"]
            g = p.grad.detach().cpu().numpy()
            g = g if g.shape == ref.shape else g[::8]
            fl = np.abs(r64).mean() + 1e-300
            e = (elem_err(g, ref, fl), elem_err(g, r64, fl), elem_err(ref, r64, fl))
            if e[1] > worst[0]:
                worst = (e[1], (k, e))
        print(f"[{mode}] {nm}: worst tensor {worst[1][0]}: vs ref32 {worst[1][1][0]:.2e} vs fp64 {worst[1][1][1]:.2e} "
              f"ref32 vs fp64 {worst[1][1][2]:.2e}")


if __name__ == "__main__":
    for mode in sys.argv[1:] or ["plain", "pin", "eager", "pin+eager"]:
        run(mode)
