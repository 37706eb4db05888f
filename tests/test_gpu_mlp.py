"""GPU parity of the fused ReLU-MLP kernels (material networks + DINO field, csrc/mlp.hip) against the eager
PyTorch expression of the same modules (which tests/test_host_logic.py pins to the reference's render_fn fixture)
and against the CPU oracle."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu

NETS = ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")


def _inputs(P, seed=0):
    g = torch.Generator().manual_seed(seed)
    F = torch.nn.functional
    pts = (torch.rand(P, 3, generator=g) - 0.5).cuda()
    nrm = F.normalize(torch.randn(P, 3, generator=g), dim=-1).cuda()
    geo = (torch.randn(P, 256, generator=g) * 0.5).cuda()
    dino = (torch.randn(P, 384, generator=g) * 0.5).cuda()
    return pts, nrm, geo, dino


def _call(net, name, pts, nrm, geo, dino):
    if name == "diffuse_albedo_network":
        return net(pts, nrm, -nrm, geo, None)
    return net(pts, nrm, None, geo, dino)


@pytest.mark.parametrize("P", [0, 1, 63, 64, 1000, 70001])
def test_material_forward_matches_eager(P):
    from pir_amd import scene, mlp_hip
    _, mats, dinonet = scene.build_networks(seed=3, device="cuda")
    pts, nrm, geo, dino = _inputs(P)
    for name in NETS:
        net = mats[name]
        assert mlp_hip.module_supported(net)
        with torch.no_grad():
            got = _call(net, name, pts, nrm, geo, dino)          # fused HIP kernel
        ref = _call(net, name, pts, nrm, geo, dino)               # parameters require grad: eager / autograd path
        assert got.shape == ref.shape and not got.requires_grad and ref.requires_grad
        if P:
            assert rel_err(got.cpu().numpy(), ref.detach().cpu().numpy()) < 2e-5, name
    got = dinonet(pts, is_training=False)
    ref = dinonet(pts, is_training=True)
    assert got.shape == (P, 384)
    if P:
        assert rel_err(got.cpu().numpy(), ref.detach().cpu().numpy()) < 2e-5


def test_material_forward_matches_oracle_and_repacks_after_update():
    import os
    from conftest import ROOT
    from oracle import pir_oracle as O
    from pir_amd import scene
    _, mats, dinonet = scene.build_networks(seed=5, device="cuda")
    pts, nrm, geo, dino = _inputs(513, seed=2)
    net = mats["specular_albedo_network"]
    cfg = scene.MODEL_CONFIG["specular_albedo_network"]
    with torch.no_grad():
        got = net(pts, nrm, None, geo, dino)
        layers = O.mlp_layers({k: v.detach().cpu() for k, v in net.state_dict().items()})
        ref = O.rendering_mlp(layers, cfg, pts.cpu(), nrm.cpu(), None, geo.cpu(), dino.cpu())
        assert rel_err(got.cpu().numpy(), ref.numpy()) < 2e-5
        # an optimizer step changes the parameters in place: the packed copy must follow
        net.lin2.weight_v.mul_(1.5)
        net.lin4.bias.add_(0.25)
        got2 = net(pts, nrm, None, geo, dino)
        layers = O.mlp_layers({k: v.detach().cpu() for k, v in net.state_dict().items()})
        ref2 = O.rendering_mlp(layers, cfg, pts.cpu(), nrm.cpu(), None, geo.cpu(), dino.cpu())
    assert rel_err(got2.cpu().numpy(), ref2.numpy()) < 2e-5
    assert rel_err(got2.cpu().numpy(), got.cpu().numpy()) > 1e-3
