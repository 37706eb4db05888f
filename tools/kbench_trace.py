#!/usr/bin/env python3
"""Micro-benchmark of the tracer kernels on the 800x800 standard view (HIP-event profiler families) -- A/B aid."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pir_amd import _lib, scene
from pir_amd.renderer.raytracer import RayTracer, raytrace_camera

sdf, _, _ = scene.build_networks(0, "cuda")
cam = scene.make_camera(800, 800, device="cuda")
tr = RayTracer()
for _ in range(2):
    raytrace_camera(cam, sdf, tr)
torch.cuda.synchronize()
L = _lib.lib()
L.pir_profile_enable(1)
reps = 5
for _ in range(reps):
    raytrace_camera(cam, sdf, tr)
torch.cuda.synchronize()
L.pir_profile_enable(0)
nk = 13
la, ms, un = (ctypes.c_long * nk)(), (ctypes.c_double * nk)(), (ctypes.c_double * nk)()
L.pir_profile_read(nk, la, ms, un)
for k, nm in ((2, "trace_step x17"), (3, "sampler"), (4, "bisect")):
    print(f"{nm:16s} {ms[k] / reps:8.3f} ms per 800x800 trace", flush=True)
