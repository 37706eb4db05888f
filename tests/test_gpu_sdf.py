"""GPU parity of the SDF-network kernels against the oracle (and through it the reference fixtures)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_from_npz, rel_err
from gpu_util import sdf_net

pytestmark = pytest.mark.gpu


def _packed(name):
    from oracle import pir_oracle as O
    from pir_amd.sdf_hip import PackedSdf
    st = state_from_npz(load_golden(name))
    layers = O.mlp_layers(st)
    pk = PackedSdf([w.cuda() for w, _ in layers], [b.cuda() for _, b in layers])
    return layers, pk


@pytest.mark.parametrize("weights", ["sdf_full.npz", "sdf_full_bumpy_weights.npz"])
def test_forward_matches_reference_fixture(weights):
    from oracle import pir_oracle as O
    layers, pk = _packed(weights)
    d = load_golden("sdf_full.npz")
    x = torch.from_numpy(d["x"])
    ref = O.sdf_forward(layers, x).detach().numpy() if weights != "sdf_full.npz" else d["forward"]
    sdf, feat = pk.forward(x.cuda(), want_feature=True)
    assert np.abs(sdf.cpu().numpy() - ref[:, 0]).max() < 2e-6 * max(1.0, np.abs(ref[:, 0]).max())
    assert rel_err(feat.cpu().numpy(), ref[:, 1:]) < 3e-6
    sdf2 = pk.forward(x.cuda())
    assert torch.equal(sdf2, sdf)


@pytest.mark.parametrize("M", [1, 63, 64, 65, 1000, 40000])
def test_get_all_matches_oracle(M):
    from oracle import pir_oracle as O
    layers, pk = _packed("sdf_full_bumpy_weights.npz")
    g = torch.Generator().manual_seed(M)
    x = (torch.rand(M, 3, generator=g) * 2 - 1)
    x[: M // 8] *= 3.0
    y, f, n = O.sdf_get_all(layers, x.clone(), is_training=False)
    sdf, feat, normal = pk.get_all(x.cuda(), want_feature=True)
    assert np.abs(sdf.cpu().numpy() - y[:, 0].numpy()).max() < 5e-6
    assert rel_err(feat.cpu().numpy(), f.numpy()) < 1e-5
    assert rel_err(normal.cpu().numpy(), n.numpy()) < 5e-6
    sdf2, none, normal2 = pk.get_all(x.cuda(), want_feature=False)
    assert none is None and torch.equal(sdf2, sdf) and torch.equal(normal2, normal)


def test_get_all_reference_fixture():
    layers, pk = _packed("sdf_full.npz")
    d = load_golden("sdf_full.npz")
    sdf, feat, normal = pk.get_all(torch.from_numpy(d["x"]).cuda())
    assert np.abs(sdf.cpu().numpy() - d["sdf"][:, 0]).max() < 5e-6
    assert rel_err(feat.cpu().numpy(), d["feature"]) < 2e-6
    assert rel_err(normal.cpu().numpy(), d["normal"]) < 5e-6


def test_empty_input():
    _, pk = _packed("sdf_full.npz")
    x = torch.zeros(0, 3, device="cuda")
    assert pk.forward(x).shape == (0,)
    s, f, n = pk.get_all(x)
    assert s.shape == (0,) and f.shape == (0, 256) and n.shape == (0, 3)


def test_packed_cache_follows_parameter_writes():
    """The packed weights are rebuilt after optimizer-style in-place updates; a write through `.data` is invisible to
    the version key and needs invalidate_packed()."""
    from pir_amd.network import invalidate_packed
    net = sdf_net("sdf_full.npz")
    x = torch.rand(100, 3).cuda() - 0.5
    a = net.sdf(x).clone()
    with torch.no_grad():
        net.lin8.bias.add_(0.25)                 # bumps the version: picked up by itself
    b = net.sdf(x)
    assert (b - a - 0.25).abs().max().item() < 1e-6
    net.lin8.bias.data.sub_(0.25)                # .data write: not visible to the key
    invalidate_packed(net)
    c = net.sdf(x)
    assert (c - a).abs().max().item() < 1e-6
