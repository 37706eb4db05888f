#!/usr/bin/env python3
"""Secondary measurements for the other BASELINE.json configs (not the headline bench line):
  config 1: 256x256 forward render, collocated light, no training       (rays/s)
  config 2: 800x800 trace + shade fwd+bwd, collocated (no shadow rays)   (rays/s)
  config 3: 800x800 visibility bake, 256 samples, forward only           (rays/s)
  config 4: the 1-GPU slice of the 8-view step, stage-3 variant: 800x800 trace + shade fwd+bwd with the offset light and
            a CACHED visibility map multiplied in (render_surface.py:634-635) -- the stage-2 variant with the live march
            is the headline bench.py line                                (rays/s)
  config 5: 2048x2048 view, L lights: visibility(128, fwd) + shade(fwd)  (ray-light pairs/s)
Every row also carries the hit count, the SDF evaluations / points the MFMA kernels processed (HIP-event profiler of the
library) and the achieved fraction of the f32 MFMA peak over the WHOLE wall time of the row.
"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pir_amd import scene
from pir_amd.render_fn import make_render_fn
from pir_amd.renderer.raytracer import (RayTracer, render_camera, render_camera_visibility, raytrace_camera,
                                        render_light_visibility, render_normal_and_color)
from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer


import ctypes
from pir_amd import _lib

PEAK = 157.3
FWD, FWD_SDF = 1049088, 918016
# algorithmic FLOP per unit of each profiled kernel family (units: points; tracer families: evaluations)
FLOP = {0: FWD_SDF, 1: 2 * FWD_SDF, 2: FWD_SDF, 3: FWD_SDF, 4: FWD_SDF, 5: 2 * FWD_SDF, 6: 2 * FWD_SDF, 9: FWD,
        10: FWD + FWD_SDF}
LAST = {}


def timed(fn, n=3, warm=1):
    """Mean wall time of fn over n runs; LAST = per-run SDF work seen by the library's profiler in those runs."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    L = _lib.lib()
    L.pir_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    L.pir_profile_enable(0)
    nk = 13
    launches, ms, units = (ctypes.c_long * nk)(), (ctypes.c_double * nk)(), (ctypes.c_double * nk)()
    L.pir_profile_read(nk, launches, ms, units)
    flop = sum(FLOP[i] * units[i] for i in FLOP) / n
    LAST.clear()
    LAST.update({"sdf_units_per_run": {"forward": (units[0] + units[9]) / n, "get_all": (units[1] + units[10]) / n,
                                       "tracer_evals": (units[2] + units[3] + units[4]) / n,
                                       "backward_points": units[5] / n},
                 "sdf_kernel_ms_per_run": round(sum(ms[i] for i in FLOP) / n, 2),
                 "mfma_TFLOPs_over_wall": round(flop / dt / 1e12, 2), "frac_of_f32_mfma_peak_over_wall": round(flop / dt / 1e12 / PEAK, 4),
                 "mfma_TFLOPs_over_sdf_kernel_time": round(flop / (sum(ms[i] for i in FLOP) / n * 1e-3) / 1e12, 2)})
    return dt


def main():
    dev = torch.device("cuda", 0)
    sdf, mats, dino = scene.build_networks(0, dev)
    tracer = RayTracer()
    render_fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    inv_s = torch.full((1, 1), scene.INV_S, device=dev)
    pl = mats["point_light_network"]
    out = []
    cam = scene.make_camera(256, 256, device=dev)
    with torch.no_grad():
        t = timed(lambda: render_camera(cam, sdf, tracer, mats, dino, render_fn, fill_holes=False, handle_edges=False,
                                        indirect_warmup=True, is_training=False))
    hits = lambda c: int(raytrace_camera(c, sdf, tracer)["convergent_mask"].sum())  # noqa: E731
    out.append(dict({"config": 1, "what": "256x256 forward, collocated", "rays_per_s": 256 * 256 / t, "ms": t * 1e3,
                     "hits": hits(cam)}, **LAST))
    cam = scene.make_camera(800, 800, device=dev)
    params = list(sdf.parameters()) + [p for k in mats for p in mats[k].parameters()]

    def step2():
        for p in params:
            p.grad = None
        r = render_camera(cam, sdf, tracer, mats, dino, render_fn, fill_holes=False, handle_edges=False,
                          indirect_warmup=True, is_training=True)
        scene.step_loss(r).backward()
    t = timed(step2)
    out.append(dict({"config": 2, "what": "800x800 trace + shade fwd+bwd, collocated", "rays_per_s": 800 * 800 / t,
                     "ms": t * 1e3, "hits": hits(cam)}, **LAST))
    with torch.no_grad():
        lp = cam.get_light_pos(pl.offset)
        t = timed(lambda: render_camera_visibility(cam, sdf, tracer, fill_holes=False, handle_edges=False,
                                                   is_training=False, inv_s=inv_s, render_visibility=True,
                                                   max_num_pts=2500, light_pos=lp))
    out.append(dict({"config": 3, "what": "800x800 visibility bake (256 samples), forward", "rays_per_s": 800 * 800 / t,
                     "ms": t * 1e3, "hits": hits(cam)}, **LAST))
    # config 4, stage-3 variant (1-GPU slice): offset light, cached visibility map (here: the bake of config 3)
    with torch.no_grad():
        vis_cached = render_camera_visibility(cam, sdf, tracer, fill_holes=False, handle_edges=False, is_training=False,
                                              inv_s=inv_s, render_visibility=True, max_num_pts=2500,
                                              light_pos=cam.get_light_pos(pl.offset))["vis"]

    def step4():
        for p in params:
            p.grad = None
        r = render_camera(cam, sdf, tracer, mats, dino, render_fn, fill_holes=False, handle_edges=False,
                          render_visibility=False, light_pos=cam.get_light_pos(pl.get_offset()), indirect_warmup=True,
                          is_training=True)
        m = r["convergent_mask"]
        loss = (r["color"] * vis_cached[..., None]).sum() + ((r["normal"][m].norm(dim=-1) - 1.0) ** 2).sum()
        loss.backward()
    t = timed(step4)
    out.append(dict({"config": 4, "what": "800x800 stage-3 step (1-GPU slice of the 8-view job): trace + shade fwd+bwd, offset "
                                          "light, cached visibility map multiplied in", "rays_per_s": 800 * 800 / t,
                     "ms": t * 1e3, "hits": hits(cam)}, **LAST))
    for o in out:
        print(json.dumps(o), flush=True)
    cam = scene.make_camera(2048, 2048, device=dev)
    import math

    def fib(n_l):   # Fibonacci sphere, radius 3
        lights = []
        for i in range(n_l):
            z = 1 - 2 * (i + 0.5) / n_l
            r = math.sqrt(1 - z * z)
            a = math.pi * (1 + 5 ** 0.5) * i
            lights.append(torch.tensor([[3 * r * math.cos(a), 3 * r * math.sin(a), 3 * z]], device=dev))
        return lights

    for n_l in [int(v) for v in os.environ.get("PIR_CONFIG5_LIGHTS", "1,16,64").split(",")]:
        lights = fib(n_l)

        def relight(lights=lights):
            with torch.no_grad():
                res = raytrace_camera(cam, sdf, tracer)
                for li, lp in enumerate(lights):
                    render_light_visibility(res, cam, lp, sdf, inv_s, n_samples=128, is_training=False)
                    render_normal_and_color(res, sdf, mats, dino, render_fn, is_training=False, light_pos=lp)
                    if li % 8 == 7:
                        print(f"[config 5] L={n_l}: light {li + 1}", file=sys.stderr, flush=True)
                return int(res["convergent_mask"].sum())
        hits = relight(lights[:1])        # warm-up (allocator state, packed weights)
        t = timed(relight, n=1, warm=0)
        out.append({"config": 5, "what": f"2048x2048 view, {n_l} lights (Fibonacci sphere r=3): trace once + per light "
                                         "visibility(128) + shade, forward", "lights": n_l, "hits": hits,
                    "ray_light_pairs_per_s": 2048 * 2048 * n_l / t, "march_samples_per_s": hits * 118 * n_l / t,
                    "ms": t * 1e3})
        out[-1].update(LAST)
        print(json.dumps(out[-1]), flush=True)



if __name__ == "__main__":
    main()
