#!/usr/bin/env python3
"""Development aid: dump dW / db / dx of the SDF backward for fixed random inputs (A/B of two library builds)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].double(), b[k].double()
        print(f"{k:12s} {tuple(x.shape)!s:14s} max abs diff {(x - y).abs().max().item():.3e}  rel to mean|ref| {((x - y).abs().max() / (y.abs().mean() + 1e-30)).item():.3e}")
    sys.exit(0)
from pir_amd import scene, sdf_hip
sdf, _, _ = scene.build_networks(seed=0, device="cuda")
torch.manual_seed(1)
M = int(os.environ.get("AB_M", "5000"))
x = (torch.rand(M, 3, device="cuda") * 2 - 1) * 0.7
gs, gf, gn = torch.randn(M, device="cuda"), torch.randn(M, 256, device="cuda") * 0.01, torch.randn(M, 3, device="cuda")
pk = sdf.packed()
out = {}
gws, gbs, gx = sdf_hip.get_all_backward(pk, x, gs, gf, gn, True)
for i, (w, b) in enumerate(zip(gws, gbs)):
    out[f"dW{i}"], out[f"db{i}"] = w.cpu(), b.cpu()
out["dx"] = gx.cpu()
gws, gbs, gx = sdf_hip.get_all_backward(pk, x, gs, None, gn, True)
for i, (w, b) in enumerate(zip(gws, gbs)):
    out[f"nofeat_dW{i}"] = w.cpu()
out["nofeat_dx"] = gx.cpu()
torch.save(out, sys.argv[1])
