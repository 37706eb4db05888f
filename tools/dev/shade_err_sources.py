"""Development aid: the tracer-free shading case with (a) fused material MLPs, (b) the eager PyTorch expression for
them -- error of the roughness / specular network gradients against the fp64 reference run."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from full800_util import full_materials, elem_err  # noqa: E402
from gpu_util import sdf_net  # noqa: E402
import test_gpu_full800 as T  # noqa: E402
from pir_amd import mlp_hip  # noqa: E402
import contextlib  # noqa: E402

d = load_golden("shade_points.npz")
for tag, ctx in (("fused", contextlib.nullcontext()), ("eager", mlp_hip.eager_grad())):
    net = sdf_net("sdf_full_bumpy_weights.npz")
    mats, dino = full_materials("cuda")
    with ctx:
        results, vis, loss = T._shade_on_fixture_points(d, net, mats, dino)
    print(tag, "loss", loss.item())
    for nm in ("specular_roughness_network", "specular_albedo_network"):
        for k, p in mats[nm].named_parameters():
            r64 = d[f"ref64.g.{nm}.{k}"]
            r32 = d[f"ref32.g.{nm}.{k}"]
            g = p.grad.cpu().numpy()
            g = g if g.shape == r64.shape else g[::8]
            fl = np.abs(r64).mean()
            print(f"  {tag} {nm[9:18]}.{k:16s} hip/64 {elem_err(g, r64, fl):.2e}  ref32/64 {elem_err(r32, r64, fl):.2e}")
