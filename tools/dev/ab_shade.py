#!/usr/bin/env python3
"""Development aid: record every SDF-backward call (inputs and outputs) and every parameter gradient of the
shading-on-fixture-points scenario of tests/test_gpu_full800.py (A/B of two library builds)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        if a[k] is None or b[k] is None:
            print(k, "None", a[k] is None, b[k] is None); continue
        x, y = a[k].double(), b[k].double()
        if x.shape != y.shape:
            print(k, "SHAPES", x.shape, y.shape); continue
        print(f"{k:34s} {tuple(x.shape)!s:14s} max abs diff {(x - y).abs().max().item():.3e}  rel to mean|ref| {((x - y).abs().max() / (y.abs().mean() + 1e-30)).item():.3e}")
    for k in ("call0.g_sdf", "call0.g_nrm", "call0.g_feat", "call0.gx"):
        x, y = a[k].double().reshape(a[k].shape[0], -1), b[k].double().reshape(a[k].shape[0], -1)
        dpt = (x - y).abs().max(dim=1).values
        top = dpt.topk(6)
        print(k, "points with diff > 1e-3 * mean|ref|:", int((dpt > 1e-3 * y.abs().mean()).sum()), "top points", top.indices.tolist(),
              ["%.2e" % v for v in top.values.tolist()], "ref there", ["%.2e" % v for v in y.abs().max(dim=1).values[top.indices].tolist()])
    sys.exit(0)
from conftest import load_golden
from gpu_util import sdf_net
from full800_util import full_materials
import test_gpu_full800 as T
from pir_amd import sdf_hip
rec = {}
orig = sdf_hip.get_all_backward
def wrapped(packed, x, g_sdf, g_feat, g_normal, need_gx, chunk_points=None):
    i = len([k for k in rec if k.endswith(".x")])
    gws, gbs, gx = orig(packed, x, g_sdf, g_feat, g_normal, need_gx, chunk_points)
    c = lambda t: None if t is None else t.detach().cpu().clone()
    rec[f"call{i}.x"], rec[f"call{i}.g_sdf"], rec[f"call{i}.g_feat"], rec[f"call{i}.g_nrm"] = c(x), c(g_sdf), c(g_feat), c(g_normal)
    rec[f"call{i}.gx"] = c(gx)
    for l in (0, 4, 8):
        rec[f"call{i}.dW{l}"] = c(gws[l])
    return gws, gbs, gx
sdf_hip.get_all_backward = wrapped
import pir_amd.sdf_autograd as sa
sa.sdf_hip.get_all_backward = wrapped
from pir_amd.sdf_hip import PackedSdf
orig_ga = PackedSdf.get_all
def ga(self, x, want_feature=True, *a, **k):
    r = orig_ga(self, x, want_feature, *a, **k)
    i = len([q for q in rec if q.endswith(".fx")])
    rec[f"fwd{i}.fx"] = x.detach().cpu().clone()
    for nm, t in zip(("sdf", "feat", "nrm"), r):
        if t is not None:
            rec[f"fwd{i}.{nm}"] = t.detach().cpu().clone()
    return r
PackedSdf.get_all = ga
d = load_golden("shade_points.npz")
net = sdf_net("sdf_full_bumpy_weights.npz")
mats, dino = full_materials("cuda")
results, vis, loss = T._shade_on_fixture_points(d, net, mats, dino)
for k in ("color", "normal", "diffuse_albedo", "specular_albedo", "specular_roughness"):
    rec["out." + k] = results[k].detach().cpu()
rec["out.vis"] = vis.detach().cpu()
for k, p in net.named_parameters():
    rec["grad.sdf." + k] = p.grad.detach().cpu()
torch.save(rec, sys.argv[1])
