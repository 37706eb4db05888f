#!/usr/bin/env python3
"""Development aid: a few launches of the fused material-MLP kernels (forward, backward) for rocprofv3 counter
passes.  usage: kprof_mlp.py [network_name]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pir_amd import scene, mlp_hip
name = sys.argv[1] if len(sys.argv) > 1 else "diffuse_albedo_network"
_, mats, dino = scene.build_networks(seed=0, device="cuda")
net = dino if name == "dino_network" else mats[name]
pk = mlp_hip.packed_for(net)
P = 131072
x = torch.randn(P, pk.k0, device="cuda")
for _ in range(2):
    with torch.no_grad():
        pk.forward(x)
    if pk.n_out <= 4:
        xg = x.clone().requires_grad_(True)
        net.zero_grad()
        mlp_hip.mlp_apply_fused(net, xg).sum().backward()
torch.cuda.synchronize()
