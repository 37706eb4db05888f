#!/usr/bin/env python3
"""Golden vectors for the FULL-SIZE material / DINO networks: runs the reference's own RenderingNetwork and
DINONetwork (src/network.py, imported from /root/reference on CPU) on weights that tests/materials_util.py can
regenerate, and writes inputs, outputs and gradients to tests/golden/materials_full.npz.

Runs ONLY in the build container.  Usage: python tools/gen_golden_materials.py [--ref /root/reference]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "tests"))
import materials_util as MU  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    sys.path.insert(0, args.ref)
    from src.network import RenderingNetwork, DINONetwork  # noqa  (torch + numpy only)
    cfg = json.load(open(os.path.join(args.ref, "configs", "model_config.json")))
    torch.set_num_threads(8)
    P = 96
    pts, nrm, geo, dino = (torch.from_numpy(a) for a in MU.inputs(P, seed=11))
    out = {"P": P, "input_seed": 11}
    for i, name in enumerate(MU.NAMES):
        net = (DINONetwork if name == "dino_network" else RenderingNetwork)(**cfg[name])
        net.load_state_dict(MU.fill_state(net.state_dict(), seed=100 + i))
        out[f"{name}.weight_seed"] = 100 + i
        out[f"{name}.weight_checksum"] = float(sum(v.double().abs().sum() for v in net.state_dict().values()))
        leaves = [t.clone().requires_grad_(True) for t in (pts, nrm, geo, dino)]
        if name == "dino_network":
            y = net(leaves[0], is_training=True)
        else:
            y = MU.call(net, name, *leaves)
        cw = torch.from_numpy(np.random.RandomState(500 + i).standard_normal(tuple(y.shape)).astype(np.float32))
        (y * cw).sum().backward()
        out[f"{name}.out"] = y.detach().numpy()
        out[f"{name}.cw"] = cw.numpy()
        for nm, t in zip(("pts", "nrm", "geo", "dino"), leaves):
            if t.grad is not None:
                out[f"{name}.g_{nm}"] = t.grad.numpy()
        for k, p in net.named_parameters():
            g = p.grad.numpy()
            out[f"{name}.g.{k}"] = g[::16] if g.ndim == 2 and g.size > 40000 else g
    path = os.path.join(ROOT, "tests", "golden", "materials_full.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
