#!/usr/bin/env python3
"""The stage-2 / stage-3 training step at the sizes the reference trains at.

    python tools/bench_train_patch.py --patch 128 --steps 20 --warmup 5 [--stage 2|3] [--device cuda|cpu]

Runs `pir_amd.train.training_step` (reference: render_surface.py:545-699) with the full-size networks of
configs/model_config.json on random crops of eight 800 x 800 views of the bumpy synthetic object, at global_step 12000:
edges on, indirect bounce on, smoothness loss on, and
  --stage 2 : --render_visibility   (live 128-sample shadow march, light offset trained; scripts/train_vis.sh:5-14)
  --stage 3 : --render_brdf         (cached visibility map multiplied in, render_surface.py:634-635; train_vis.sh:17-26)
Patch 64 is the reference's stage-2 size, 128 its stage-3 size, 256 the roughness warm-up size.

Prints one JSON line: it/s, rays/s, ms/step, mean hits and edge pixels per step.  Under
`rocprofv3 --kernel-trace` the timed region is bracketed by two `erfinv` marker kernels (nothing else in the step
uses that op), so tools/train_patch_report.py can compute Sum(kernel time) / wall = GPU-busy fraction and
launches/step for exactly the timed steps.  --device cpu times the same training_step through the eager CPU
expressions (the port of the reference path that the host-logic tests pin against the reference fixtures).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_state(device, patch, stage, n_views=8, res=800, bumpy=True, handle_edges=True, smoothness=1e-4):
    from pir_amd import scene, train
    from pir_amd.render_fn import make_render_fn
    from pir_amd.renderer.raytracer import RayTracer
    from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer
    dev = torch.device(device)
    sdf, nets, dino = scene.build_networks(seed=0, device="cpu")
    if bumpy:      # perturb lin*.weight_v by N(0, 0.02 std) with seed 1 -> silhouette edges, self-shadowing
        g = torch.Generator().manual_seed(1)
        with torch.no_grad():
            for name, p in sdf.named_parameters():
                if name.endswith("weight_v"):
                    p.add_(torch.randn(p.shape, generator=g) * 0.02 * p.std())
    sdf, dino = sdf.to(dev), dino.to(dev)
    nets = {k: v.to(dev) for k, v in nets.items()}
    cfg = train.TrainConfig(num_iters=50001, patch_size=patch, render_visibility=(stage == 2),
                            render_brdf=(stage == 3), smoothness_weight=smoothness, handle_edges=handle_edges)
    lr = dict(scene.MODEL_CONFIG["lr"], dino_network=1e-3)
    sdf_opt, color_opts, dino_opt = train.make_optimizers(sdf, nets, dino, lr, render_visibility=(stage == 2))
    cams = [scene.make_camera(res, res, view=v, n_views=n_views, device=dev) for v in range(n_views)]
    g = torch.Generator().manual_seed(2)
    gt = torch.rand(n_views, res, res, 3, generator=g).to(dev)
    grads = [torch.rand(res, res, generator=g).to(dev) for _ in range(n_views)]
    vis = torch.rand(n_views, res, res, generator=g).to(dev) if stage == 3 else None
    state = train.TrainState(
        args=cfg, cameras=cams, gt_images=gt, grad_images=grads, sdf_network=sdf, color_network_dict=nets,
        dino_network=dino, raytracer=RayTracer(),
        render_fn=make_render_fn(GGXColocatedRenderer(use_cuda=dev.type == "cuda"), smoothness_weight=smoothness),
        inv_s=torch.full((1, 1), scene.INV_S, device=dev), sdf_optimizer=sdf_opt, color_optimizer_dict=color_opts,
        dino_optimizer=dino_opt, vis_images=vis)
    return state


def object_crops(res, patch, n, seed):
    """Upper-left corners of n random patches that overlap the object's silhouette disc (radius ~ res/3 around the
    image centre on the standard view): the reference draws corners uniformly (raytracer.py:358-383); most of its
    dataset images are filled by the object, the synthetic view is not, so the draw is restricted to the disc's
    bounding box to keep the per-step work representative."""
    rng = np.random.RandomState(seed)
    lo, hi = int(res / 2 - res / 3 - patch / 2), int(res / 2 + res / 3 - patch / 2)
    lo, hi = max(0, lo), min(res - patch - 1, hi)
    return [(int(rng.randint(lo, hi + 1)), int(rng.randint(lo, hi + 1))) for _ in range(n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stage", type=int, default=2, choices=[2, 3])
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--global-step", type=int, default=12000)
    ap.add_argument("--no-edges", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    dev = torch.device(args.device)
    state = build_state(dev, args.patch, args.stage, handle_edges=not args.no_edges)
    from pir_amd import train
    from pir_amd.renderer.raytracer import Camera
    # the timed steps always see the same crops and views, whatever the warm-up length (warm-up cycles through them too)
    corners = object_crops(800, args.patch, args.steps, seed=3)
    it = {"i": 0}
    orig_crop = Camera.crop_region

    def crop_region(self, trgt_W, trgt_H, center_crop=False, ul_corner=None, images=None):
        c = corners[it["i"] % len(corners)]
        return orig_crop(self, trgt_W, trgt_H, ul_corner=c, images=images)
    Camera.crop_region = crop_region
    np.random.seed(4)
    torch.manual_seed(5)
    sync = torch.cuda.synchronize if dev.type == "cuda" else (lambda: None)
    marker = torch.full((1,), 0.5, device=dev)
    hits = edges = 0
    for i in range(args.warmup):
        it["i"] = i
        train.training_step(state, args.global_step)
    sync()
    np.random.seed(4)
    torch.manual_seed(5)
    if dev.type == "cuda":
        torch.erfinv(marker)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        it["i"] = i
        out = train.training_step(state, args.global_step)
        if dev.type == "cpu":
            print(f"[train_patch] cpu step {i + 1}/{args.steps} ({time.perf_counter() - t0:.0f} s)", file=sys.stderr,
                  flush=True)
        hits = hits + out["convergent_mask"].sum()          # device-side accumulation: no extra host sync
        edges = edges + out["results"]["edge_mask"].sum()
    sync()
    dt = time.perf_counter() - t0
    if dev.type == "cuda":
        torch.erfinv(marker)
        sync()
    hits, edges = int(hits), int(edges)
    rays = args.patch * args.patch * args.steps
    print(json.dumps({"tool": "bench_train_patch", "device": str(dev), "stage": args.stage, "patch": args.patch,
                      "steps": args.steps, "warmup": args.warmup, "global_step": args.global_step,
                      "handle_edges": not args.no_edges, "indirect": args.global_step >= 10000,
                      "it_per_s": round(args.steps / dt, 3), "rays_per_s": round(rays / dt, 1),
                      "ms_per_step": round(dt / args.steps * 1e3, 2), "hits_per_step": round(hits / args.steps, 1),
                      "edge_pixels_per_step": round(edges / args.steps, 1),
                      "threads": torch.get_num_threads() if dev.type == "cpu" else None}), flush=True)


if __name__ == "__main__":
    main()
