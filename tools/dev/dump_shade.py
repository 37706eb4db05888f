"""Development aid: run the tracer-free shading case of tests/test_gpu_full800.py on the GPU and dump every HIP
gradient to gpurun_out/shade_hip.npz for offline error analysis."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from full800_util import full_materials  # noqa: E402
from gpu_util import sdf_net  # noqa: E402
import test_gpu_full800 as T  # noqa: E402

d = load_golden("shade_points.npz")
net = sdf_net("sdf_full_bumpy_weights.npz")
mats, dino = full_materials("cuda")
results, vis, loss = T._shade_on_fixture_points(d, net, mats, dino)
out = {"vis": vis.detach().cpu().numpy(), "color": results["color"].detach().cpu().numpy()}
for k, p in net.named_parameters():
    out["g.sdf." + k] = p.grad.cpu().numpy()
for nm in ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
    for k, p in mats[nm].named_parameters():
        out[f"g.{nm}.{k}"] = p.grad.cpu().numpy()
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "shade_hip.npz"), **out)
print("dumped", len(out))
