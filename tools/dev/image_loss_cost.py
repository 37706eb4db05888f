#!/usr/bin/env python3
"""Development aid: wall time of the image losses (L1 + pyramid L2 + masked SSIM, forward + backward) of one training step."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pir_amd.models.image_losses import PyramidL2Loss, ssim_loss_fn

for P in (64, 128, 256):
    torch.manual_seed(0)
    pred = torch.rand(1, 3, P, P, device="cuda", requires_grad=True)
    gt = torch.rand(1, 3, P, P, device="cuda")
    mask = (torch.rand(1, 1, P, P, device="cuda") > 0.02)
    pyr = PyramidL2Loss(use_cuda=True)

    def step():
        pred.grad = None
        loss = torch.abs(pred - gt).mean() + pyr(pred, gt) + ssim_loss_fn(pred, gt, mask)
        loss.backward()
        return loss
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    print(f"patch {P}: image losses fwd+bwd {1e3 * (time.perf_counter() - t0) / n:.2f} ms per step", flush=True)
