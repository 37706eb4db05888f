"""Checkpoint compatibility with the reference (src/loader.py:11-48, render_surface.py:716-722).

Checkpoints are plain dictionaries of state dicts with the reference's keys; they are read with
`weights_only=True` (tensors only) and mapped to the device the networks live on.  The dataset reader of
src/loader.py (imageio / cv2 image decoding) is outside the rendering path and not provided.
"""
import os

import torch


def path2step(path):
    """'ckpt_5000.pth' / 'dino_5000.pth' -> 5000."""
    return int(os.path.basename(path)[5:-4])


def _load(path, like):
    dev = next(like.parameters()).device
    return torch.load(path, map_location=dev, weights_only=True)


def load_from_neus(neus_ckpt_fpath, sdf_network, color_network_dict, deviation_network):
    """NeuS stage-1 checkpoint -> SDF, diffuse albedo and variance networks (src/loader.py:11-17)."""
    ckpt = _load(neus_ckpt_fpath, sdf_network)
    sdf_network.load_state_dict(ckpt["sdf_network_fine"])
    color_network_dict["diffuse_albedo_network"].load_state_dict(ckpt["color_network_fine"])
    deviation_network.load_state_dict(ckpt["variance_network_fine"])


def load_dino_ckpt(dino_ckpt_fpaths, dino_network):
    """Latest 'dino_<step>.pth' -> DINO field; returns its step or -1 (src/loader.py:20-31)."""
    if len(dino_ckpt_fpaths) == 0:
        return -1
    path = sorted(dino_ckpt_fpaths, key=path2step)[-1]
    dino_network.load_state_dict(_load(path, dino_network)["dino_network"])
    return path2step(path)


def load_pir_ckpt(ckpt_fpaths, sdf_network, color_network_dict):
    """Latest 'ckpt_<step>.pth' -> SDF + every network of color_network_dict; returns its step or -1
    (src/loader.py:34-48)."""
    if len(ckpt_fpaths) == 0:
        return -1
    path = sorted(ckpt_fpaths, key=path2step)[-1]
    ckpt = _load(path, sdf_network)
    sdf_network.load_state_dict(ckpt["sdf_network"])
    for name in list(color_network_dict.keys()):
        color_network_dict[name].load_state_dict(ckpt[name])
    return path2step(path)


def save_pir_ckpt(out_dir, global_step, sdf_network, color_network_dict):
    """The dictionary render_surface.py:716-722 writes every 5000 steps."""
    path = os.path.join(out_dir, f"ckpt_{global_step}.pth")
    torch.save(dict([("sdf_network", sdf_network.state_dict())] +
                    [(k, v.state_dict()) for k, v in color_network_dict.items()]), path)
    return path
