#!/usr/bin/env python3
"""Development aid: the stage-2 soak run step by step -- every loss term, and at the first non-finite value which
outputs / gradients / parameters carry it.   python tools/dev/soak_diag.py [--device cuda|cpu] [--steps 30]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--device", default="cuda")
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--patch", type=int, default=64)
ap.add_argument("--stage", type=int, default=2)
ap.add_argument("--threads", type=int, default=0)
args = ap.parse_args()
if args.threads:
    torch.set_num_threads(args.threads)
import bench_train_patch as B
from pir_amd import scene, train
from pir_amd.renderer.raytracer import render_camera, render_camera_visibility
dev = torch.device(args.device)
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
                          indirect_warmup=True)
        img = r["color"] * vis[..., None]
        gts.append(torch.where(img > 1.0, img.clamp(min=1.0) ** 0.05, img))
        vis_maps.append(vis)
state.gt_images = torch.stack(gts)
state.vis_images = torch.stack(vis_maps)
state.grad_images = [torch.zeros(400, 400, device=dev) for _ in state.cameras]
_, fresh, _ = scene.build_networks(seed=7, device="cpu")
for k in ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
    state.color_network_dict[k].load_state_dict(fresh[k].state_dict())
np.random.seed(11)
torch.manual_seed(12)


def named_params():
    yield from (("sdf." + n, p) for n, p in state.sdf_network.named_parameters())
    for k, net in state.color_network_dict.items():
        yield from ((k + "." + n, p) for n, p in net.named_parameters())


for step in range(1, args.steps + 1):
    out = train.training_step(state, 12000 + step)
    terms = {k: float(out[k]) for k in ("loss", "img_loss", "img_l1_loss", "img_ssim_loss", "eik_loss") if k in out}
    gmax = {}
    for n, p in named_params():
        if p.grad is not None:
            top = n.split(".")[0]
            gmax[top] = max(gmax.get(top, 0.0), float(p.grad.abs().max()))
    print(step, {k: round(v, 6) for k, v in terms.items()}, "hits", int(out["convergent_mask"].sum()),
          "gradmax", {k: float(f"{v:.3g}") for k, v in gmax.items()},
          "light_offset", [round(float(x), 5) for x in pl.get_offset().detach().flatten()], flush=True)
    bad = not all(np.isfinite(v) for v in terms.values())
    badp = [n for n, p in named_params() if not bool(torch.isfinite(p).all())]
    badg = [n for n, p in named_params() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    if bad or badp or badg:
        print("NON-FINITE at step", step, "params:", badp[:8], "grads:", badg[:8])
        for k, v in out["results"].items():
            if torch.is_tensor(v) and v.is_floating_point() and not bool(torch.isfinite(v).all()):
                print("   results[%s]: %d non-finite of %d" % (k, int((~torch.isfinite(v)).sum()), v.numel()))
        break
