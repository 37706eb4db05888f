#!/usr/bin/env python3
"""Development aid: a few launches of the SDF kernels for rocprofv3 counter passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pir_amd.sdf_hip import PackedSdf
torch.manual_seed(0)
shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
pk = PackedSdf([torch.randn(s, device="cuda") * 0.05 for s in shapes], [torch.randn(s[0], device="cuda") * 0.05 for s in shapes])
x = torch.rand(1 << 21, 3, device="cuda") * 2 - 1
for _ in range(2):
    pk.forward(x)
    pk.get_all(x, want_feature=False)
torch.cuda.synchronize()
