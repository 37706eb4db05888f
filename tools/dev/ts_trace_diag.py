#!/usr/bin/env python3
"""Development aid: the pinned training step of tests/train_step_util.py with the loaded library; dumps the outputs of
every tracer call and of the edge walk, to compare two builds (which discrete decision moved?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import train_step_util
from pir_amd.renderer import raytracer as rt

calls = []
orig_fwd = rt.RayTracer.forward


def rec_fwd(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask, *a, **k):
    out = orig_fwd(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask, *a, **k)
    calls.append({"ray_o": ray_o, "ray_d": ray_d, "mask": out["convergent_mask"], "points": out["points"],
                  "distance": out["distance"]})
    return out


rt.RayTracer.forward = rec_fwd
orig_loc = rt.locate_edge_points
edge = {}


def rec_loc(*a, **k):
    r = orig_loc(*a, **k)
    edge.update({kk: v for kk, v in r.items() if torch.is_tensor(v)})
    return r


rt.locate_edge_points = rec_loc
out = train_step_util.run("cuda", tol=1e-4, grad_tol=1.0, exact=False, mat_grad_tol=1.0, adjudicate=False, pin_points=True)
res = {}
for i, c in enumerate(calls):
    for k, v in c.items():
        res[f"call{i}.{k}"] = v.detach().cpu().numpy()
for k, v in edge.items():
    res["edge." + k] = v.detach().cpu().numpy()
r = out["results"]
for k in ("color", "normal", "indirect_color", "indirect_blend_coef", "specular_roughness", "specular_albedo", "points"):
    if k in r:
        res["res." + k] = r[k].detach().cpu().numpy()
np.savez(sys.argv[1], **res)
print("wrote", sys.argv[1], len(calls), "tracer calls")
