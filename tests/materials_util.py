"""Deterministic full-size material / DINO network weights for tests/golden/materials_full.npz.

The fixture stores inputs, outputs and gradients produced by the REFERENCE modules (tools/gen_golden_materials.py)
but not the 1.7 M weights: both the generator and the tests fill every state-dict tensor from numpy's frozen
legacy generator (RandomState), whose streams do not change between numpy versions."""
import numpy as np
import torch

NAMES = ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network", "dino_network")


def fill_state(state_dict, seed):
    """New state dict with the same keys/shapes: weight_v ~ N(0, 2/in), weight_g = |v| * U(0.8, 1.2), bias ~ N(0, 0.05)."""
    rs = np.random.RandomState(seed)
    out = {}
    for k in sorted(state_dict.keys()):
        shp = tuple(state_dict[k].shape)
        if k.endswith("weight_v"):
            out[k] = (rs.standard_normal(shp) * np.sqrt(2.0 / shp[1])).astype(np.float32)
        elif k.endswith("bias"):
            out[k] = (rs.standard_normal(shp) * 0.05).astype(np.float32)
    for k in sorted(state_dict.keys()):
        if k.endswith("weight_g"):
            v = out[k.replace("weight_g", "weight_v")]
            out[k] = (np.linalg.norm(v, axis=1, keepdims=True) * rs.uniform(0.8, 1.2, size=(v.shape[0], 1))).astype(np.float32)
    return {k: torch.from_numpy(v) for k, v in out.items()}


def inputs(P, seed):
    rs = np.random.RandomState(seed)
    pts = (rs.uniform(-0.6, 0.6, size=(P, 3))).astype(np.float32)
    nrm = rs.standard_normal((P, 3)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    geo = (rs.standard_normal((P, 256)) * 0.5).astype(np.float32)
    dino = (rs.standard_normal((P, 384)) * 0.5).astype(np.float32)
    return pts, nrm, geo, dino


def call(net, name, pts, nrm, geo, dino):
    if name == "diffuse_albedo_network":
        return net(pts, nrm, -nrm, geo, None)
    if name == "dino_network":
        return net(pts, is_training=False)
    return net(pts, nrm, None, geo, dino)
