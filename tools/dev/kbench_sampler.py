#!/usr/bin/env python3
"""Development aid: tracer kernel time per trace on silhouette crops of the bumpy training scene (the dense sampler has
work there, unlike on centre crops of the smooth scene)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from pir_amd import _lib
from pir_amd.renderer.raytracer import raytrace_camera
from bench_train_patch import build_state, object_crops

L = _lib.lib()
for P in (64, 128, 256):
    st = build_state("cuda", P, 3)
    corners = object_crops(800, P, 8, seed=3)
    cams = [st.cameras[i % len(st.cameras)].crop_region(P, P, ul_corner=c)[0] for i, c in enumerate(corners)]
    for cam in cams:
        raytrace_camera(cam, st.sdf_network, st.raytracer)
    torch.cuda.synchronize()
    L.pir_profile_enable(1)
    reps = 5
    for _ in range(reps):
        for cam in cams:
            raytrace_camera(cam, st.sdf_network, st.raytracer)
    torch.cuda.synchronize()
    L.pir_profile_enable(0)
    nk = 13
    la, ms, un = (ctypes.c_long * nk)(), (ctypes.c_double * nk)(), (ctypes.c_double * nk)()
    L.pir_profile_read(nk, la, ms, un)
    n = reps * len(cams)
    hits = sum(int(raytrace_camera(c, st.sdf_network, st.raytracer)["convergent_mask"].sum()) for c in cams) / len(cams)
    print(f"patch {P}: " + ", ".join(f"{nm} {ms[k] / n:.3f} ms" for k, nm in ((2, "trace_step x17"), (3, "sampler"), (4, "bisect")))
          + f"  mean hits {hits:.0f}  evals per trace: trace_step {un[2] / n:.0f}, sampler {un[3] / n:.0f}, bisect {un[4] / n:.0f}"
          + f"  -> sampler {un[3] * 918016 / (ms[3] * 1e-3) / 1e12:.1f} TFLOP/s", flush=True)
