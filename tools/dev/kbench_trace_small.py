#!/usr/bin/env python3
"""Development aid: tracer kernel time per trace at training-patch sizes (64 x 64 and 128 x 128 crops of the standard view)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pir_amd import _lib, scene
from pir_amd.renderer.raytracer import RayTracer, raytrace_camera

sdf, _, _ = scene.build_networks(0, "cuda")
tr = RayTracer()
L = _lib.lib()
for P in (64, 128, 256):
    cam, _ = scene.make_camera(800, 800, device="cuda").crop_region(P, P, ul_corner=(400 - P // 2, 330 - P // 2))
    for _ in range(3):
        raytrace_camera(cam, sdf, tr)
    torch.cuda.synchronize()
    L.pir_profile_enable(1)
    reps = 20
    for _ in range(reps):
        raytrace_camera(cam, sdf, tr)
    torch.cuda.synchronize()
    L.pir_profile_enable(0)
    nk = 13
    la, ms, un = (ctypes.c_long * nk)(), (ctypes.c_double * nk)(), (ctypes.c_double * nk)()
    L.pir_profile_read(nk, la, ms, un)
    print(f"patch {P}: " + ", ".join(f"{nm} {ms[k] / reps:.3f} ms" for k, nm in ((2, "trace_step x17"), (3, "sampler"), (4, "bisect")))
          + f"  hits {int(raytrace_camera(cam, sdf, tr)['convergent_mask'].sum())}", flush=True)
