#!/usr/bin/env python3
"""Development aid: run the shading of tests/golden/shade_points.npz with the loaded library and dump per-point
quantities (normals, d loss / d normal, visibility) + two SDF gradient tensors, to compare two builds point by point."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from conftest import load_golden
from gpu_util import sdf_net
from full800_util import full_materials
import test_gpu_full800 as T

d = load_golden("shade_points.npz")
net = sdf_net("sdf_full_bumpy_weights.npz")
mats, dino = full_materials("cuda")
hooks = {}
import pir_amd.render_fn as RF
orig = RF.make_render_fn


def patched(*a, **k):
    fn = orig(*a, **k)

    def wrapped(interior_mask, color_network_dict, ray_o, ray_d, points, normals, *rest, **kw):
        normals.retain_grad(); points.retain_grad()
        hooks["normals"], hooks["points"] = normals, points
        return fn(interior_mask, color_network_dict, ray_o, ray_d, points, normals, *rest, **kw)
    return wrapped


RF.make_render_fn = patched
results, vis, loss = T._shade_on_fixture_points(d, net, mats, dino)
out = {"vis": vis.detach().cpu().numpy(), "color": results["color"].detach().cpu().numpy(),
       "normals_in": hooks["normals"].detach().cpu().numpy(), "g_normals_in": hooks["normals"].grad.cpu().numpy(),
       "g_points_in": hooks["points"].grad.cpu().numpy()}
for k, p in net.named_parameters():
    if k in ("lin1.weight_v", "lin8.weight_v", "lin0.bias"):
        out["g." + k] = p.grad.cpu().numpy()
for k, p in mats["diffuse_albedo_network"].named_parameters():
    if k in ("lin0.bias", "lin1.bias"):
        out["gd." + k] = p.grad.cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1])
