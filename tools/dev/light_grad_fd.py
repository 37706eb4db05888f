#!/usr/bin/env python3
"""Development aid: finite-difference check of d loss / d light offset through the whole stage-2 training step
(trace, live 128-sample visibility, shading, image losses) on fixed crops of the soak scene."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import bench_train_patch as B
from pir_amd import train
from pir_amd.renderer.raytracer import Camera, render_camera

dev = torch.device("cuda")
P = 64
state = B.build_state(dev, P, 2, n_views=8, res=400)
pl = state.color_network_dict["point_light_network"]
gts = []
with torch.no_grad():
    for cam in state.cameras:
        lp = cam.get_light_pos(pl.get_offset())
        r = render_camera(cam, state.sdf_network, state.raytracer, state.color_network_dict, state.dino_network,
                          state.render_fn, fill_holes=False, handle_edges=False, is_training=False, light_pos=lp,
                          indirect_warmup=True, inv_s=state.inv_s, render_visibility=True, max_num_pts=200000)
        img = r["color"] * r["vis"][..., None]
        gts.append(torch.where(img > 1.0, img.clamp(min=1.0) ** 0.05, img))
state.gt_images = torch.stack(gts)
state.grad_images = [torch.zeros(400, 400, device=dev) for _ in state.cameras]
for g in state.sdf_optimizer.param_groups + [g for o in state.color_optimizer_dict.values() for g in o.param_groups]:
    g["lr"] = 0.0
base = pl.light_offset.weight.detach().clone() + torch.tensor([[0.05, -0.04, 0.03]], device=dev)
orig = Camera.crop_region
eik = torch.empty(P * P // 2, 3, device=dev).uniform_(-1.0, 1.0)
for corner in ((168, 168), (60, 150), (150, 250)):
    Camera.crop_region = lambda self, trgt_W, trgt_H, center_crop=False, ul_corner=None, images=None, c=corner: \
        orig(self, trgt_W, trgt_H, ul_corner=c, images=images)

    def loss_at(off):
        with torch.no_grad():
            pl.light_offset.weight.copy_(off)
        np.random.seed(3)
        out = train.training_step(state, 12001, eik_points=eik)
        return out["loss"], pl.light_offset.weight.grad.detach().clone().flatten(), int(out["convergent_mask"].sum())

    l0, g0, hits = loss_at(base)
    print(f"crop {corner}: hits {hits} loss {l0:.6f} autograd d loss / d offset = {[round(float(v), 5) for v in g0]}")
    for h in (2e-3, 5e-4):
        fd = []
        for c in range(3):
            e = torch.zeros(1, 3, device=dev); e[0, c] = h
            fd.append((loss_at(base + e)[0] - loss_at(base - e)[0]) / (2 * h))
        print(f"    central differences h={h}: {[round(float(v), 5) for v in fd]}")
