"""Development aid: per-point error of the HIP GGX backward against an fp64 evaluation of the PyTorch formula, next
to the error of the fp32 PyTorch formula (what the reference runs), on ggx.npz's 4096 points and on random points."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer  # noqa: E402

ggx = GGXColocatedRenderer(use_cuda=True)
d = load_golden("ggx.npz")
names = ("distance", "normal", "lightdir", "viewdir", "kd", "ks", "alpha")
base = [torch.from_numpy(d[n]).cuda() for n in names]
g = torch.Generator().manual_seed(4)
w = torch.rand(base[1].shape, generator=g).cuda() + 0.5


def run(fn, dtype):
    leaves = [t.to(dtype).clone().requires_grad_(True) for t in base]
    light = torch.tensor(135.0, device="cuda", dtype=dtype, requires_grad=True)
    out = fn(light, *leaves)
    (out["rgb"] * w.to(dtype)).sum().backward()
    return [out["rgb"].detach().double()] + [t.grad.double() for t in leaves] + [light.grad.double()]


hip = run(lambda light, *a: ggx(light, *a), torch.float32)
t32 = run(lambda light, *a: ggx._forward_torch(light, *a, True), torch.float32)
t64 = run(lambda light, *a: ggx._forward_torch(light, *a, True), torch.float64)
for i, nm in enumerate(("rgb",) + tuple("g_" + n for n in names) + ("g_light",)):
    ref = t64[i]
    fl = ref.abs().mean()
    eh = ((hip[i] - ref).abs() / (ref.abs() + fl))
    et = ((t32[i] - ref).abs() / (ref.abs() + fl))
    print(f"{nm:12s} hip: max {eh.max().item():.2e} p99.9 {eh.flatten().quantile(0.999).item():.2e} sum-err "
          f"{abs((hip[i].sum() - ref.sum()).item()) / (ref.abs().sum().item() + 1e-300):.2e} | torch fp32: max {et.max().item():.2e} p99.9 "
          f"{et.flatten().quantile(0.999).item():.2e} sum-err {abs((t32[i].sum() - ref.sum()).item()) / (ref.abs().sum().item() + 1e-300):.2e}")
