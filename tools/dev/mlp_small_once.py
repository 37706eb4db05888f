#!/usr/bin/env python3
"""Development aid: 50 fused material-network forward+backward calls at one point count (run under rocprofv3 --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pir_amd import scene, mlp_hip
P = int(sys.argv[1]) if len(sys.argv) > 1 else 3400
_, mats, dino = scene.build_networks(seed=0, device="cuda")
net = mats["diffuse_albedo_network"]
pk = mlp_hip.packed_for(net)
x = torch.randn(P, pk.k0, device="cuda").requires_grad_(True)
for _ in range(50):
    net.zero_grad(); x.grad = None
    mlp_hip.mlp_apply_fused(net, x).sum().backward()
torch.cuda.synchronize()
