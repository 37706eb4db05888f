#!/usr/bin/env python3
"""Development aid: dump get_all forward / backward results of the loaded library for a fixed random problem
(compare two builds: PIR_HIP_LIB=... python tools/dev/bwd_dump.py out.npz)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pir_amd import sdf_hip
from pir_amd.sdf_hip import PackedSdf

torch.manual_seed(0)
shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
ws = [torch.randn(s, device="cuda") * 0.05 for s in shapes]
bs = [torch.randn(s[0], device="cuda") * 0.05 for s in shapes]
pk = PackedSdf(ws, bs)
M = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
x = torch.rand(M, 3, device="cuda") * 2 - 1
gs, gn = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
gf = torch.randn(M, 256, device="cuda") * 0.1
out = {"x": x.cpu().numpy(), "gs": gs.cpu().numpy(), "gn": gn.cpu().numpy(), "gf": gf.cpu().numpy()}
for i in range(9):
    out[f"W{i}"], out[f"b{i}"] = ws[i].cpu().numpy(), bs[i].cpu().numpy()
y, f, n = pk.get_all(x, want_feature=True)
out["y"], out["n"] = y.cpu().numpy(), n.cpu().numpy()
for tag, feat in (("nofeat", None), ("feat", gf)):
    gws, gbs, gx = sdf_hip.get_all_backward(pk, x, gs, feat, gn, True)
    out[tag + ".gx"] = gx.cpu().numpy()
    for i, (a, b) in enumerate(zip(gws, gbs)):
        out[f"{tag}.gw{i}"], out[f"{tag}.gb{i}"] = a.cpu().numpy(), b.cpu().numpy()
np.savez(sys.argv[1], **out)
print("wrote", sys.argv[1])
