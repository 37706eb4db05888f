#!/usr/bin/env python3
"""Development aid: where does a training crop of the soak scene differ from the teacher image it is fitted to?
Per region (interior hits / edge pixels / background) the mean |pred - gt|, with and without the indirect term."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--patch", type=int, default=64)
ap.add_argument("--stage", type=int, default=2)
ap.add_argument("--teacher-indirect", action="store_true")
args = ap.parse_args()
import bench_train_patch as B
from pir_amd import train
from pir_amd.renderer.raytracer import Camera, render_camera, render_camera_visibility
dev = torch.device("cuda")
state = B.build_state(dev, args.patch, args.stage, n_views=8, res=400)
pl = state.color_network_dict["point_light_network"]
gts, vis_maps = [], []
with torch.no_grad():
    for cam in state.cameras:
        lp = cam.get_light_pos(pl.get_offset())
        vis = render_camera_visibility(cam, state.sdf_network, state.raytracer, fill_holes=False, handle_edges=False,
                                       is_training=False, inv_s=state.inv_s, render_visibility=True, max_num_pts=200000,
                                       light_pos=lp)["vis"]
        r = render_camera(cam, state.sdf_network, state.raytracer, state.color_network_dict, state.dino_network,
                          state.render_fn, fill_holes=False, handle_edges=False, is_training=False, light_pos=lp,
                          indirect_warmup=not args.teacher_indirect, inv_s=state.inv_s,
                          render_visibility=args.stage == 2, max_num_pts=200000)
        img = r["color"] * (r["vis"] if args.stage == 2 else vis)[..., None]
        if args.teacher_indirect:
            img = img + r["indirect_color"] * r["indirect_blend_coef"]
        gts.append(torch.where(img > 1.0, img.clamp(min=1.0) ** 0.05, img))
        vis_maps.append(vis)
state.gt_images = torch.stack(gts)
state.vis_images = torch.stack(vis_maps)
state.grad_images = [torch.zeros(400, 400, device=dev) for _ in state.cameras]
cap = {}
orig = Camera.crop_region


def crop_region(self, trgt_W, trgt_H, center_crop=False, ul_corner=None, images=None):
    out = orig(self, trgt_W, trgt_H, center_crop=center_crop, ul_corner=ul_corner, images=images)
    cap["crop"] = out[1]
    return out


Camera.crop_region = crop_region
np.random.seed(11); torch.manual_seed(12)
for g in state.sdf_optimizer.param_groups + [g for o in state.color_optimizer_dict.values() for g in o.param_groups]:
    g["lr"] = 0.0        # look, do not train
for step in range(1, 13):
    out = train.training_step(state, 12000 + step)
    res = out["results"]
    gt = cap["crop"]["rgb"].to(dev)
    conv, edge = res["convergent_mask"], res["edge_mask"]
    direct = res["color"] * (res["vis"][..., None] if args.stage == 2 else cap["crop"]["vis"].to(dev)[..., None])
    ind = res["indirect_color"] * res["indirect_blend_coef"]
    comp = lambda x: torch.where(x > 1.0, x.clamp(min=1.0) ** 0.05, x)
    pred, pred_noind = comp(direct + ind).detach(), comp(direct).detach()
    regions = {"interior": conv & ~edge, "edge": edge, "background": ~conv & ~edge}
    msg = []
    for nm, m in regions.items():
        if int(m.sum()):
            msg.append(f"{nm} n={int(m.sum())} |pred-gt| {float((pred - gt)[m].abs().mean()):.4f} "
                       f"(no indirect {float((pred_noind - gt)[m].abs().mean()):.4f}, max {float((pred - gt)[m].abs().max()):.3f}, "
                       f"mean |indirect| {float(ind[m].abs().mean().detach()):.4f}, mean gt {float(gt[m].mean()):.3f})")
    print(step, f"loss {out['loss']:.4f} l1 {out['img_l1_loss']:.4f}", " | ".join(msg), flush=True)
