#!/usr/bin/env python3
"""Profiled program for the MFMA-utilisation / effective-clock PMC pass (tools/profile_mfma.sh): long dispatches of
get_all (4 M points, ~60 ms each) and of the SDF backward (bwd_sweep_kernel + wgrad256_kernel on 262144-point chunks),
repeated so that the chip is at its loaded clock when the last ones run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pir_amd import sdf_hip
from pir_amd.sdf_hip import PackedSdf
torch.manual_seed(0)
shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
pk = PackedSdf([torch.randn(s, device="cuda") * 0.05 for s in shapes], [torch.randn(s[0], device="cuda") * 0.05 for s in shapes])
M = 1 << 22
x = torch.rand(M, 3, device="cuda") * 2 - 1
gs, gn = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
for _ in range(int(os.environ.get("KPROF_ITERS", "6"))):
    pk.get_all(x, want_feature=False)
    sdf_hip.get_all_backward(pk, x, gs, None, gn, True)
torch.cuda.synchronize()
