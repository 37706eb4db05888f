"""Stage-2/3 training-step glue around the rendering path (reference: render_surface.py:495-699).

The reference runs this as inline script code on module globals; here the same logic is two functions on an
explicit `TrainState`:

  train_visibility(state, out_dir)   visibility bake + the `vis_images.npy` cache ([n_views, H, W] fp32)
  training_step(state, global_step)  one iteration: view + crop selection, render_camera, image / eikonal /
                                     roughness-range / smoothness / light-offset losses, backward, and the
                                     optimizer schedule of the warm-up phases
Random choices use the same generators in the same order as the reference (np.random for the view and the crop,
torch for the eikonal samples), so a seeded step is reproducible against it.
"""
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from .models.image_losses import PyramidL2Loss, ssim_loss_fn
from .renderer.raytracer import render_camera, render_camera_visibility


@dataclass
class TrainConfig:
    """The command-line arguments the step reads (src/parser.py defaults)."""
    num_iters: int = 50001
    patch_size: int = 128
    eik_weight: float = 0.1
    ssim_weight: float = 1.0
    roughrange_weight: float = 0.1
    smoothness_weight: float = 0.0001
    roughness_warmup_step: int = 2000
    indirect_warmup_step: int = 10000
    render_visibility_step: int = 4001
    gamma_pred: bool = False
    render_visibility: bool = False
    render_brdf: bool = False
    fill_holes: bool = False
    handle_edges: bool = True


@dataclass
class TrainState:
    args: TrainConfig
    cameras: list
    gt_images: torch.Tensor                 # [n_views, H, W, 3]
    grad_images: List[torch.Tensor]         # per view [H, W] image-gradient magnitude
    sdf_network: torch.nn.Module
    color_network_dict: Dict[str, torch.nn.Module]
    dino_network: torch.nn.Module
    raytracer: torch.nn.Module
    render_fn: object
    inv_s: torch.Tensor
    sdf_optimizer: torch.optim.Optimizer
    color_optimizer_dict: Dict[str, torch.optim.Optimizer]
    dino_optimizer: Optional[torch.optim.Optimizer] = None
    vis_images: Optional[torch.Tensor] = None
    pyramidl2_loss_fn: Optional[torch.nn.Module] = None
    grad_hook: Optional[object] = None      # called between backward and the optimizer steps (data-parallel all-reduce)
    last: dict = field(default_factory=dict)


def make_optimizers(sdf_network, color_network_dict, dino_network, lr, render_visibility):
    """render_surface.py:213-231: Adam per network; the light offset is only trainable with --render_visibility."""
    sdf_opt = torch.optim.Adam(sdf_network.parameters(), lr=lr["sdf_network"])
    pl = color_network_dict["point_light_network"]
    groups = [{"params": [pl.light], "lr": 2e-5}]
    if render_visibility:
        groups.append({"params": list(pl.light_offset.parameters()), "lr": 1e-4})
    color_opts = {}
    for name, net in color_network_dict.items():
        color_opts[name] = torch.optim.Adam(groups) if name == "point_light_network" else \
            torch.optim.Adam(net.parameters(), lr=lr[name])
    dino_opt = torch.optim.Adam(dino_network.parameters(), lr=lr.get("dino_network", 1e-3))
    return sdf_opt, color_opts, dino_opt


def train_visibility(state, out_dir, fill_holes=False, handle_edges=True):
    """render_surface.py:495-520: per-view visibility maps (256-sample march), cached as vis_images.npy."""
    fpath = os.path.join(out_dir, "vis_images.npy")
    dev = state.inv_s.device
    if os.path.isfile(fpath):
        return torch.from_numpy(np.load(fpath, allow_pickle=False)).to(dev)
    maps = []
    pl = state.color_network_dict["point_light_network"]
    for cam in state.cameras:
        res = render_camera_visibility(cam, state.sdf_network, state.raytracer, fill_holes=fill_holes,
                                       handle_edges=handle_edges, is_training=False, inv_s=state.inv_s,
                                       render_visibility=True, max_num_pts=2500,
                                       light_pos=cam.get_light_pos(pl.offset))
        maps.append(res["vis"])
    vis = torch.stack(maps, dim=0)
    np.save(fpath, vis.detach().cpu().numpy())
    return vis


def training_step(state, global_step, eik_points=None):
    """One iteration of render_surface.py:545-699.  Returns the dictionary of loss terms (floats detached in
    `state.last`).  `eik_points` overrides the uniform eikonal samples (tests)."""
    a = state.args
    nets = state.color_network_dict
    pl = nets["point_light_network"]
    dev = state.inv_s.device
    state.sdf_optimizer.zero_grad()
    for opt in state.color_optimizer_dict.values():
        opt.zero_grad()
    if state.dino_optimizer is not None:
        state.dino_optimizer.zero_grad()

    roughness_warmup = global_step < a.roughness_warmup_step
    indirect_warmup = global_step < a.indirect_warmup_step
    sdf_optim = not roughness_warmup
    idx = np.random.randint(0, state.gt_images.shape[0])
    image_width = a.patch_size
    if a.render_visibility and roughness_warmup:
        image_width = 256
    render_shadow = a.render_visibility and global_step >= a.render_visibility_step
    images = {"rgb": state.gt_images[idx], "grad": state.grad_images[idx]}
    if a.render_brdf:
        images["vis"] = state.vis_images[idx]
    camera_crop, crop = state.cameras[idx].crop_region(trgt_W=image_width, trgt_H=image_width, images=images)
    gt_color, gt_grad_img, gt_vis = crop["rgb"], crop["grad"], crop.get("vis", None)

    if a.render_visibility:
        light_pos = camera_crop.get_light_pos(pl.offset if render_shadow else torch.zeros((1, 3), device=dev))
    elif a.render_brdf:
        light_pos = camera_crop.get_light_pos(pl.get_offset())
    else:
        light_pos = None

    results = render_camera(camera_crop, state.sdf_network, state.raytracer, nets, state.dino_network,
                            state.render_fn, fill_holes=a.fill_holes, handle_edges=a.handle_edges, is_training=True,
                            inv_s=state.inv_s, render_visibility=render_shadow, light_pos=light_pos,
                            roughness_warmup=roughness_warmup, indirect_warmup=indirect_warmup)
    convergent_mask = results["convergent_mask"].clone()
    mask = convergent_mask | results["edge_mask"] if a.handle_edges else convergent_mask

    zero = torch.zeros(1, device=dev)
    img_loss = img_l1 = img_l2 = img_ssim = roughrange_loss = smoothness_loss = zero
    # eikonal term on uniform samples of the unit cube
    if eik_points is None:
        eik_points = torch.empty(camera_crop.H * camera_crop.W // 2, 3, device=dev).float().uniform_(-1.0, 1.0)
    eik_grad = state.sdf_network.gradient(eik_points).view(-1, 3)
    eik_cnt = eik_grad.shape[0]
    eik_loss = ((eik_grad.norm(dim=-1) - 1) ** 2).sum()
    if mask.any():
        pred = results["color"].permute(2, 0, 1).unsqueeze(0)
        if a.gamma_pred:
            pred = torch.pow(pred + 1e-6, 1.0 / 2.2)
        if render_shadow:
            pred = pred * results["vis"].expand(pred.shape)
        elif a.render_brdf:
            pred = pred * gt_vis.to(dev).expand(pred.shape)
        if not indirect_warmup:
            pred = pred + (results["indirect_color"] * results["indirect_blend_coef"]).permute(2, 0, 1).unsqueeze(0)
        # over-exposed pixels are compressed (x^0.05) rather than clipped
        over = pred > 1.0
        pred = torch.where(over, pred.clamp(min=1.0) ** 0.05, pred)
        gt_img = gt_color.permute(2, 0, 1).unsqueeze(0).to(dev)
        gt_grad = gt_grad_img.to(dev)[convergent_mask].reshape(-1)
        if state.pyramidl2_loss_fn is None:
            state.pyramidl2_loss_fn = PyramidL2Loss(use_cuda=dev.type == "cuda")
        img_l1 = torch.abs(pred - gt_img).mean()
        img_l2 = state.pyramidl2_loss_fn(pred, gt_img)
        img_ssim = a.ssim_weight * ssim_loss_fn(pred, gt_img, mask.unsqueeze(0).unsqueeze(0))
        img_loss = img_l1 + img_l2 + img_ssim

        n = results["normal"][mask]
        eik_cnt += n.shape[0]
        eik_loss = eik_loss + ((n.norm(dim=-1) - 1) ** 2).sum()
        if "edge_pos_neg_normal" in results:
            n = results["edge_pos_neg_normal"]
            eik_cnt += n.shape[0]
            eik_loss = eik_loss + ((n.norm(dim=-1) - 1) ** 2).sum()

        rough = results["specular_roughness"][mask]
        rough = rough[rough > 0.5]
        if rough.numel() > 0:
            roughrange_loss = (rough - 0.5).mean() * a.roughrange_weight
        if a.smoothness_weight > 0.0:
            sg = results["specular_grads"][convergent_mask]
            rg = results["roughness_grads"][convergent_mask]
            smoothness_loss = ((sg.norm(dim=-1) / 1.732 + rg.norm(dim=-1)) * (-gt_grad).exp()).mean() * a.smoothness_weight

    eik_loss = eik_loss / eik_cnt * a.eik_weight
    loss = img_loss + eik_loss + roughrange_loss + smoothness_loss
    if a.render_visibility:
        loss = loss + torch.abs(pl.offset[0][2]) * 0.5
    loss.backward()
    if state.grad_hook is not None:
        state.grad_hook()

    if sdf_optim:
        state.sdf_optimizer.step()
    if roughness_warmup:
        state.color_optimizer_dict["diffuse_albedo_network"].step()
        state.color_optimizer_dict["specular_albedo_network"].step()
    elif render_shadow:
        state.color_optimizer_dict["point_light_network"].step()
    else:
        for opt in state.color_optimizer_dict.values():
            opt.step()

    f = lambda t: float(t.detach().reshape(-1)[0]) if torch.is_tensor(t) else float(t)  # noqa: E731
    state.last = dict(loss=f(loss), img_loss=f(img_loss), img_l1_loss=f(img_l1), img_l2_loss=f(img_l2),
                      img_ssim_loss=f(img_ssim), eik_loss=f(eik_loss), roughrange_loss=f(roughrange_loss),
                      smoothness_loss=f(smoothness_loss), idx=idx, camera_crop=camera_crop, results=results,
                      convergent_mask=convergent_mask, render_shadow=render_shadow)
    # the float() reads above synchronised the stream: reading the accumulated tracer status costs no extra wait
    if hasattr(state.raytracer, "check_status"):
        state.raytracer.check_status()
    return state.last
