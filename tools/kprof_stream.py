#!/usr/bin/env python3
"""A few launches of every HBM-bound kernel of the path at P points (default 4 M), for rocprofv3 passes
(--kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE): GGX forward / backward, light-march sample positions,
compositing forward / backward, shading inputs forward / backward, frequency encoding forward / backward.
    python tools/kprof_stream.py [P] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pir_amd import visibility_hip, shade_hip, embed_hip  # noqa: E402
from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 22
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
F = torch.nn.functional
torch.manual_seed(0)
g = lambda *s: torch.rand(*s, device="cuda")  # noqa: E731
ggx = GGXColocatedRenderer(use_cuda=True)
nrm, ld, vd = (F.normalize(torch.randn(P, 3, device="cuda"), dim=-1) for _ in range(3))
dist, kd, ks, al = g(P, 1) * 3 + 0.5, g(P, 3), g(P, 3), g(P, 1) * 0.8 + 0.01
light = torch.tensor(135.0, device="cuda")
leaves = [x.clone().requires_grad_(True) for x in (dist, nrm, ld, vd, kd, ks, al)]
w = torch.rand(P, 3, device="cuda")
n_s = 118
bs = max(1, P // n_s)
inv_s = torch.tensor([[20.0855]], device="cuda")
sdf = (torch.rand(bs * n_s, device="cuda") * 0.2 - 0.02).requires_grad_(True)
nr = torch.randn(bs * n_s, 3, device="cuda").requires_grad_(True)
d = F.normalize(torch.randn(bs, 3, device="cuda"), dim=-1)
Lg = torch.rand(bs, 1, device="cuda") + 2.5
lp = torch.tensor([[0.3, 0.2, -2.9]], device="cuda")
pts = (torch.rand(P, 3, device="cuda") - 0.5)
sd = (torch.rand(P, 1, device="cuda") * 1e-4).requires_grad_(True)
gn = torch.randn(P, 3, device="cuda").requires_grad_(True)
ray_d = F.normalize(torch.randn(P, 3, device="cuda"), dim=-1)
ray_o = torch.zeros(P, 3, device="cuda") + torch.tensor([0.0, 0.0, -3.0], device="cuda")
x_e = (torch.rand(P, 3, device="cuda") * 2 - 1).requires_grad_(True)
for _ in range(reps):
    out = ggx(light, *leaves)["rgb"]
    torch.autograd.grad(out, leaves, w)
    with torch.no_grad():
        visibility_hip.march_points(lp, d, Lg, 128)
    vis = visibility_hip.composite(sdf, nr, d, Lg, inv_s, 128)
    torch.autograd.grad(vis, [sdf, nr], torch.ones_like(vis))
    p2, nh, ldir, dd = shade_hip.shade_inputs(pts, sd, gn, ray_d, ray_o, lp, True)
    torch.autograd.grad([p2, nh, ldir, dd], [sd, gn], [w, w, w, torch.ones_like(dd)])
    e = embed_hip.embed(x_e, 10)
    torch.autograd.grad(e, x_e, torch.ones_like(e))
torch.cuda.synchronize()
print("P", P, "march rays", bs, "samples", bs * n_s)
