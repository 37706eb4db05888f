import sys, os
sys.path.insert(0, "/root/repo")
import torch
from pir_amd import scene, mlp_hip
_, mats, _ = scene.build_networks(seed=4, device="cuda")
P = int(sys.argv[1]); name = sys.argv[2] if len(sys.argv) > 2 else "diffuse_albedo_network"
net = mats[name]
pk = mlp_hip.packed_for(net)
torch.manual_seed(0)
x = torch.randn(P, pk.k0, device="cuda") * 0.5
w = torch.randn(P, pk.n_out, device="cuda")
res = []
for fn in (mlp_hip.mlp_apply_fused, mlp_hip.mlp_apply_eager):
    xg = x.clone().requires_grad_(True); net.zero_grad()
    (fn(net, xg) * w).sum().backward()
    res.append((xg.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
d = (res[0][0] - res[1][0]).abs().max(-1).values / res[1][0].abs().max()
bad = (d > 1e-4).nonzero().flatten()
print("gx bad rows", bad.numel(), bad[:10].tolist(), bad[-5:].tolist())
if bad.numel():
    r = int(bad[0]); dc = (res[0][0][r] - res[1][0][r]).abs(); print("row", r, "bad cols", (dc > 1e-5).nonzero().flatten()[:20].tolist())
for k in res[0][1]:
    a, b = res[0][1][k], res[1][1][k]
    e = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
    if e > 1e-4: print(k, e)
