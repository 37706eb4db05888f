#!/usr/bin/env python3
"""Development aid: kernel time of the fused material-network forward and backward at training-patch point counts."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pir_amd import _lib, scene, mlp_hip

L = _lib.lib()
_, mats, dino = scene.build_networks(seed=0, device="cuda")
net = mats["diffuse_albedo_network"]
pk = mlp_hip.packed_for(net)
for P in (1000, 3400, 4096, 8192, 8193, 13000, 16384):
    x = torch.randn(P, pk.k0, device="cuda").requires_grad_(True)
    def step():
        net.zero_grad(); x.grad = None
        mlp_hip.mlp_apply_fused(net, x).sum().backward()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L.pir_profile_enable(1)
    reps = 20
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    L.pir_profile_enable(0)
    nk = 13
    la, ms, un = (ctypes.c_long * nk)(), (ctypes.c_double * nk)(), (ctypes.c_double * nk)()
    L.pir_profile_read(nk, la, ms, un)
    print(f"P={P}: forward {1e3 * ms[11] / reps:.1f} us, backward (sweep + wgrad + head) {1e3 * ms[12] / reps:.1f} us", flush=True)
