#!/usr/bin/env python3
"""Development aid: forward / get_all latency on short calls (the hit points of a training patch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from pir_amd import scene

sdf, _, _ = scene.build_networks(0, "cuda")
pk = sdf.packed()


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for M in (1000, 3000, 6000, 12000, 16384, 20000, 40000):
    x = torch.rand(M, 3, device="cuda") - 0.5
    print(f"M={M:6d}: forward {t(lambda: pk.forward(x)):7.1f} us, get_all {t(lambda: pk.get_all(x, want_feature=False)):7.1f} us, "
          f"get_all+features {t(lambda: pk.get_all(x, want_feature=True)):7.1f} us", flush=True)
