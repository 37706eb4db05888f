#!/usr/bin/env python3
"""Soak run of the drop-in training step: a student scene (re-initialised material networks) is fitted to images
rendered from a teacher scene, for a few hundred iterations of pir_amd.train.training_step at the reference's patch
size (stage 3: cached visibility maps, edges on, indirect bounce on).  Logs, per window of steps: the mean loss terms,
iterations / s, hit count, allocator state and the tracer status -- the things a one-step parity test cannot show
(losses fall, memory is flat, nothing degenerates).  One JSON line per window.

  python tools/train_soak.py --steps 300 --patch 128 [--res 400] [--stage 3]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--window", type=int, default=25)
    ap.add_argument("--patch", type=int, default=128)
    ap.add_argument("--res", type=int, default=400)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--stage", type=int, default=3, choices=[2, 3])
    ap.add_argument("--student-sdf", default="teacher", choices=["teacher", "smooth"])
    ap.add_argument("--device", default="cuda", help="cpu: the eager CPU expressions of the same step, as a cross-check")
    args = ap.parse_args()
    import bench_train_patch as B
    from pir_amd import scene, train
    from pir_amd.renderer.raytracer import render_camera, render_camera_visibility
    dev = torch.device(args.device)
    gpu = dev.type == "cuda"
    chunk = 200000 if gpu else 2048        # points per shading / visibility chunk of the teacher renders
    state = B.build_state(dev, args.patch, args.stage, n_views=args.views, res=args.res)
    pl = state.color_network_dict["point_light_network"]
    # teacher images: the scene as built (bumpy SDF, seed-0 materials), every view, with its visibility map
    gts, vis_maps = [], []
    with torch.no_grad():
        for cam in state.cameras:
            lp = cam.get_light_pos(pl.get_offset())
            vis = None
            if args.stage == 3:
                vis = render_camera_visibility(cam, state.sdf_network, state.raytracer, fill_holes=False,
                                               handle_edges=False, is_training=False, inv_s=state.inv_s,
                                               render_visibility=True, max_num_pts=chunk, light_pos=lp)["vis"]
            # stage 2 marches the shadow ray live with 128 samples, the baked map has 256 (raytracer.py:1278-1310); with
            # inv_s = 20 the two discretisations differ by 10-20 % on lit points, so each stage gets the teacher image
            # of its own march
            r = render_camera(cam, state.sdf_network, state.raytracer, state.color_network_dict, state.dino_network,
                              state.render_fn, fill_holes=False, handle_edges=False, is_training=False, light_pos=lp,
                              indirect_warmup=True, inv_s=state.inv_s, render_visibility=args.stage == 2,
                              max_num_pts=chunk)
            img = r["color"] * (r["vis"] if args.stage == 2 else vis)[..., None]
            gts.append(torch.where(img > 1.0, img.clamp(min=1.0) ** 0.05, img))
            if vis is not None:
                vis_maps.append(vis)
    state.gt_images = torch.stack(gts)
    if vis_maps:
        state.vis_images = torch.stack(vis_maps)
    state.grad_images = [torch.zeros(args.res, args.res, device=dev) for _ in state.cameras]
    if args.stage == 3:
        # student: material networks re-initialised (another seed), same geometry to start from
        _, fresh, _ = scene.build_networks(seed=7, device="cpu")
        for k in ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
            state.color_network_dict[k].load_state_dict(fresh[k].state_dict())
    else:
        # stage 2 steps only the SDF and the light (render_surface.py:687-695; the material networks are frozen there):
        # the student keeps the teacher's materials and starts from a displaced light offset
        with torch.no_grad():
            pl.light_offset.weight.add_(torch.tensor([[0.05, -0.04, 0.03]], device=dev))
        if args.student_sdf == "smooth":
            # ... and from the un-perturbed geometry (the teacher is the bumpy variant): something for the SDF to learn
            smooth, _, _ = scene.build_networks(seed=0, device="cpu")
            state.sdf_network.load_state_dict(smooth.state_dict())
    np.random.seed(11)
    torch.manual_seed(12)
    if gpu:
        torch.cuda.reset_peak_memory_stats()
    params = list(state.sdf_network.parameters()) + [p for n in state.color_network_dict.values() for p in n.parameters()]
    acc, t0, hits, nan_steps = {}, time.perf_counter(), 0, 0
    for step in range(1, args.steps + 1):
        out = train.training_step(state, 12000 + step)
        for k in ("loss", "img_loss", "img_l1_loss", "img_ssim_loss", "eik_loss"):
            acc[k] = acc.get(k, 0.0) + (out[k] if np.isfinite(out[k]) else 0.0)
        hits += int(out["convergent_mask"].sum())
        if not np.isfinite(out["loss"]):
            # The reference's SSIM term is the mean over the pixels whose 11 x 11 window lies inside the mask
            # (image_losses.py:154-158): a crop that barely touches the object leaves none, the mean of nothing is NaN
            # and so is that step's loss VALUE, while every gradient stays finite.  Same here; count such steps, leave
            # their NaN terms out of the window sums, and stop only if a parameter stops being finite.
            nan_steps += 1
            if not all(bool(torch.isfinite(p).all()) for p in params):
                raise SystemExit(f"non-finite parameter at step {step}")
        if step % args.window == 0:
            if gpu:
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"tool": "train_soak", "stage": args.stage, "patch": args.patch, "res": args.res,
                              "steps": [step - args.window + 1, step], "it_per_s": round(args.window / dt, 2),
                              **{k: round(v / args.window, 6) for k, v in acc.items()},
                              "hits_per_step": round(hits / args.window, 1), "steps_with_empty_ssim_mask": nan_steps,
                              "light_offset": [round(float(v), 5) for v in pl.get_offset().flatten()],
                              "device": dev.type, "student_sdf": args.student_sdf,
                              "mem_allocated_MB": round(torch.cuda.memory_allocated() / 2 ** 20, 1) if gpu else None,
                              "mem_reserved_MB": round(torch.cuda.memory_reserved() / 2 ** 20, 1) if gpu else None,
                              "mem_peak_MB": round(torch.cuda.max_memory_allocated() / 2 ** 20, 1) if gpu else None}),
                  flush=True)
            acc, t0, hits, nan_steps = {}, time.perf_counter(), 0, 0
    if not all(bool(torch.isfinite(p).all()) for p in params):
        raise SystemExit("non-finite parameter at the end")
    if hasattr(state.raytracer, "check_status"):
        state.raytracer.check_status()


if __name__ == "__main__":
    main()
