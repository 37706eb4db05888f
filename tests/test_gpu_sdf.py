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


def test_shared_parameter_node_survives_interleaved_backward_calls():
    """All kernel calls of a step share ONE autograd node for the network parameters (SDFNetwork.theta, WNLinear.
    shared_weight).  Unlike torch's weight-norm nodes it holds no autograd-saved tensors, so the pattern
    forward A, forward B, backward A, backward B -- legal with the reference's per-call weight_norm -- keeps working;
    the summed gradients equal one backward of the summed loss; an optimizer step makes the next call rebuild it."""
    from pir_amd import scene
    g = torch.Generator().manual_seed(3)
    xa, xb = [((torch.rand(700, 3, generator=g) * 2 - 1) * 0.8).cuda() for _ in range(2)]
    ca, cb = [torch.randn(700, 3, generator=g).cuda() for _ in range(2)]

    def loss_of(net, x, c):
        s, f, n = net.get_all(x.clone(), is_training=True)
        return (s.sum() + (n * c).sum() + 0.01 * f.sum())
    net = sdf_net("sdf_full_bumpy_weights.npz")
    la, lb = loss_of(net, xa, ca), loss_of(net, xb, cb)
    assert net._theta is not None
    la.backward()
    lb.backward()                       # the shared node is traversed a second time
    g_two = {k: p.grad.clone() for k, p in net.named_parameters()}
    ref = sdf_net("sdf_full_bumpy_weights.npz")
    (loss_of(ref, xa, ca) + loss_of(ref, xb, cb)).backward()
    for k, p in ref.named_parameters():
        err = ((g_two[k] - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)).item()
        assert err < 8e-6, (k, err)
    # a parameter update invalidates the shared node and the packed weights
    theta_before = net._theta
    s0 = net.get_all(xa.clone(), is_training=True)[0].detach().clone()
    opt = torch.optim.SGD(net.parameters(), lr=1e-2)
    opt.step()
    s1 = net.get_all(xa.clone(), is_training=True)[0].detach()
    assert net._theta is not theta_before and (s1 - s0).abs().max().item() > 0
    # modifying a parameter in place between forward and backward is refused, as autograd refuses it for saved tensors
    l2 = loss_of(net, xa, ca)
    with torch.no_grad():
        net.lin3.weight_v.mul_(1.0)
    with pytest.raises(RuntimeError):
        l2.backward()
    # material networks: same pattern through WNLinear.shared_weight
    _, mats, _ = scene.build_networks(seed=0, device="cuda")
    m = mats["specular_roughness_network"]
    pts, nrm = xa[:256], torch.nn.functional.normalize(ca[:256], dim=-1)
    geo, dino = torch.randn(256, 256, generator=g).cuda(), torch.randn(256, 384, generator=g).cuda()
    o1 = m(pts, nrm, None, geo, dino).sum()
    o2 = (m(pts * 0.5, nrm, None, geo, dino) ** 2).sum()
    o1.backward()
    o2.backward()
    got = {k: p.grad.clone() for k, p in m.named_parameters()}
    m.zero_grad()
    (m(pts, nrm, None, geo, dino).sum() + (m(pts * 0.5, nrm, None, geo, dino) ** 2).sum()).backward()
    for k, p in m.named_parameters():
        assert ((got[k] - p.grad).abs().max() / (p.grad.abs().max() + 1e-30)).item() < 8e-6, k


def test_backward_is_deterministic_and_linear_in_the_upstream_gradient():
    """Two size-independent properties of pir_sdf_get_all_backward on 100 000 points (several tiles per workgroup, split-K
    slabs in the weight-gradient GEMM): the same call twice gives bit-identical gradients (ordered slab reductions, no
    float atomics), and the gradients are linear in (g_sdf, g_feat, g_normal): backward(a g1 + b g2) = a backward(g1)
    + b backward(g2) to fp32 round-off."""
    from pir_amd import sdf_hip
    _, pk = _packed("sdf_full_bumpy_weights.npz")
    M = 100_000
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.rand(M, 3, device="cuda", generator=g) * 2 - 1) * 0.8
    ups = [(torch.randn(M, device="cuda", generator=g), torch.randn(M, 256, device="cuda", generator=g) * 0.1,
            torch.randn(M, 3, device="cuda", generator=g)) for _ in range(2)]

    def run(gs, gf, gn):
        gws, gbs, gx = sdf_hip.get_all_backward(pk, x, gs, gf, gn, True)
        return [t.clone() for t in gws] + [t.clone() for t in gbs] + [gx.clone()]
    r1, r1_again, r2 = run(*ups[0]), run(*ups[0]), run(*ups[1])
    assert all(torch.equal(a, b) for a, b in zip(r1, r1_again))
    a, b = 0.75, -1.5
    mix = run(*[a * u + b * v for u, v in zip(*ups)])
    worst = 0.0
    for m, p, q in zip(mix, r1, r2):
        want = a * p + b * q
        scale = max(float(p.abs().max()), float(q.abs().max()), 1e-30)
        worst = max(worst, float((m - want).abs().max()) / scale)
    print(f"\nlinearity of the backward in its upstream gradient: largest deviation {worst:.2e} of the tensor's scale")
    assert worst < 1e-4          # 8.9e-6 on the box it was written on
