#!/usr/bin/env python3
"""Micro-benchmarks of single kernels (HIP-event timed) -- development aid, not the headline bench."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pir_amd.sdf_hip import PackedSdf


def timeit(fn, n=5, warm=2):
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
    torch.manual_seed(0)
    shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
    ws = [torch.randn(s, device="cuda") * 0.05 for s in shapes]
    bs = [torch.randn(s[0], device="cuda") * 0.05 for s in shapes]
    pk = PackedSdf(ws, bs)
    for M in (65536, 1 << 20, 1 << 22):
        x = torch.rand(M, 3, device="cuda") * 2 - 1
        t = timeit(lambda: pk.forward(x))
        print(f"forward value  M={M}: {t:.3f} ms  {M/t/1e3:.1f} Mpts/s  {M*918016/t/1e9:.1f} TFLOP/s")
        t = timeit(lambda: pk.forward(x, want_feature=True))
        print(f"forward +feat  M={M}: {t:.3f} ms  {M/t/1e3:.1f} Mpts/s  {M*1049088/t/1e9:.1f} TFLOP/s")
        t = timeit(lambda: pk.get_all(x, want_feature=False))
        print(f"get_all nofeat M={M}: {t:.3f} ms  {M/t/1e3:.1f} Mpts/s  {M*2*918016/t/1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
