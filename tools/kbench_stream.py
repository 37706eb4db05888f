#!/usr/bin/env python3
"""Achieved HBM GB/s of the streaming kernels (GGX shader, light-march compositing) -- development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pir_amd import visibility_hip
from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer


def timeit(fn, n=7, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


def main():
    F = torch.nn.functional
    ggx = GGXColocatedRenderer(use_cuda=True)
    for P in (1 << 20, 1 << 22, 1 << 24):
        g = lambda *s: torch.rand(*s, device="cuda")  # noqa: E731
        nrm, ld, vd = (F.normalize(torch.randn(P, 3, device="cuda"), dim=-1) for _ in range(3))
        dist, kd, ks, al = g(P, 1) * 3 + 0.5, g(P, 3), g(P, 3), g(P, 1) * 0.8 + 0.01
        light = torch.tensor(135.0, device="cuda")
        with torch.no_grad():
            t = timeit(lambda: ggx(light, dist, nrm, ld, vd, kd, ks, al))
        print(f"ggx forward  P={P}: {t:.3f} ms  {P*104/t/1e6:.0f} GB/s (104 B/pt algorithmic)")
        leaves = [x.clone().requires_grad_(True) for x in (dist, nrm, ld, vd, kd, ks, al)]
        out = ggx(light, *leaves)["rgb"]
        w = torch.rand_like(out)
        t = timeit(lambda: torch.autograd.grad(out, leaves, w, retain_graph=True))
        print(f"ggx backward P={P}: {t:.3f} ms  {P*172/t/1e6:.0f} GB/s (172 B/pt algorithmic; includes autograd glue)")
    inv_s = torch.tensor([[20.0855]], device="cuda")
    for bs in (1 << 16, 1 << 18):
        n = 118
        sdf = torch.rand(bs * n, device="cuda") * 0.2 - 0.02
        nr = torch.randn(bs * n, 3, device="cuda")
        d = F.normalize(torch.randn(bs, 3, device="cuda"), dim=-1)
        L = torch.rand(bs, 1, device="cuda") + 2.5
        with torch.no_grad():
            t = timeit(lambda: visibility_hip.composite(sdf, nr, d, L, inv_s, 128))
        print(f"march composite fwd bs={bs}: {t:.3f} ms  {bs*n*16/t/1e6:.0f} GB/s (16 B/sample)")
        s_, n_ = sdf.clone().requires_grad_(True), nr.clone().requires_grad_(True)
        vis = visibility_hip.composite(s_, n_, d, L, inv_s, 128)
        gv = torch.rand_like(vis)
        t = timeit(lambda: torch.autograd.grad(vis, [s_, n_], gv, retain_graph=True))
        print(f"march composite bwd bs={bs}: {t:.3f} ms  {bs*n*32/t/1e6:.0f} GB/s (32 B/sample)")
        lp = torch.tensor([[0.3, 0.2, -2.9]], device="cuda")
        with torch.no_grad():
            t = timeit(lambda: visibility_hip.march_points(lp, d, L, 128))
        print(f"march points      bs={bs}: {t:.3f} ms  {bs*n*12/t/1e6:.0f} GB/s (12 B/sample)")


if __name__ == "__main__":
    main()
