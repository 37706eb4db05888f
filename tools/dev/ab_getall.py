#!/usr/bin/env python3
"""Development aid: get_all (no feature / feature) at a multi-tile size, dumped for an A/B of two library builds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        x, y = a[k].double(), b[k].double()
        d = (x - y).abs().reshape(x.shape[0], -1).max(dim=1).values
        print(f"{k:12s} max abs diff {d.max().item():.3e}  points differing {int((d > 0).sum())} of {d.numel()}  first {d.nonzero()[:5, 0].tolist()}")
    sys.exit(0)
from pir_amd import scene
sdf, _, _ = scene.build_networks(seed=0, device="cuda")
torch.manual_seed(1)
M = int(os.environ.get("AB_M", "300000"))
x = (torch.rand(M, 3, device="cuda") * 2 - 1) * 0.7
pk = sdf.packed()
out = {}
s, _, n = pk.get_all(x, want_feature=False)
out["sdf"], out["normal"] = s.cpu(), n.cpu()
s, f, n = pk.get_all(x, want_feature=True)
out["sdf_f"], out["normal_f"] = s.cpu(), n.cpu()
s2, _, n2 = pk.get_all(x, want_feature=False)
out["normal_again"] = n2.cpu()
torch.save(out, sys.argv[1])
