#!/usr/bin/env python3
"""Micro-benchmark of the fused material / DINO MLP kernels against the eager rocBLAS expression -- development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pir_amd import scene, mlp_hip


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
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 306000
    _, mats, dino = scene.build_networks(seed=0, device="cuda")
    nets = {k: mats[k] for k in ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")}
    nets["dino_network"] = dino
    for name, net in nets.items():
        pk = mlp_hip.packed_for(net)
        x = torch.randn(P, pk.k0, device="cuda")
        flop = 2 * sum(l.weight_v.numel() for l in mlp_hip.module_layers(net))
        with torch.no_grad():
            t_hip = timeit(lambda: pk.forward(x))
            t_eager = timeit(lambda: mlp_hip.mlp_apply(net, x))
        print(f"{name:28s} P={P} k0={pk.k0}: fused {t_hip:7.3f} ms ({flop*P/t_hip/1e9:6.1f} TFLOP/s)   "
              f"eager {t_eager:7.3f} ms ({flop*P/t_eager/1e9:6.1f} TFLOP/s)", flush=True)
        if "--train" in sys.argv and pk.plan.has_bwd:
            xg = x.clone().requires_grad_(True)
            def step(fn):
                net.zero_grad(); xg.grad = None
                fn(net, xg).sum().backward()
            t_hip = timeit(lambda: step(mlp_hip.mlp_apply_fused))
            t_eager = timeit(lambda: step(mlp_hip.mlp_apply_eager))
            print(f"{'':28s} fwd+bwd: fused {t_hip:7.3f} ms ({3*flop*P/t_hip/1e9:6.1f} TFLOP/s)   "
                  f"eager {t_eager:7.3f} ms ({3*flop*P/t_eager/1e9:6.1f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
