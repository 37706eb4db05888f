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


@pytest.mark.parametrize("P", [0, 1, 33, 63, 64, 1000, 8193, 70001])   # up to 8192 points: 32-point tiles
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


@pytest.mark.parametrize("P", [1, 33, 64, 1000])
def test_material_backward_matches_eager(P):
    """Fused backward (recompute + sweeps + split-K weight gradients) against autograd on the eager expression:
    gradients of every parameter (through weight norm) and of every input."""
    from pir_amd import scene, mlp_hip
    _, mats, _ = scene.build_networks(seed=4, device="cuda")
    pts, nrm, geo, dino = _inputs(P, seed=P)
    g = torch.Generator().manual_seed(7)
    for name in NETS:
        net = mats[name]
        res = []
        for mode in ("fused", "eager"):
            leaves = [t.clone().requires_grad_(True) for t in (pts, nrm, geo, dino)]
            net.zero_grad()
            if mode == "eager":
                with mlp_hip.eager_grad():
                    out = _call(net, name, *leaves)
            else:
                out = _call(net, name, *leaves)
            if mode == "fused":
                w = torch.randn(out.shape, generator=g).cuda()
            (out * w).sum().backward()
            res.append((out.detach(), [l.grad for l in leaves], {k: p.grad.clone() for k, p in net.named_parameters()}))
        (o0, gl0, gp0), (o1, gl1, gp1) = res
        assert rel_err(o0.cpu().numpy(), o1.cpu().numpy()) < 2e-6
        for a, b, nm in zip(gl0, gl1, ("points", "normals", "geo_f", "dino_f")):
            if b is None:
                assert a is None or float(a.abs().max()) == 0.0, nm
                continue
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5, (name, nm)
        for k in gp0:
            assert rel_err(gp0[k].cpu().numpy(), gp1[k].cpu().numpy()) < 2e-5, (name, k)


def test_short_call_tiles_agree_with_full_tiles():
    """Calls of up to 8192 points run 32-point tiles in one operand set, longer ones 64-point tiles in two
    (csrc/mlp.hip, PIR_MLP_HALF).  The arithmetic per point is the same: the first 5000 rows of a 9000-point call
    (upstream gradient zero on the other rows) must give the outputs and input gradients of the 5000-point call bit
    for bit; the parameter gradients are sums over rows in another split, so they agree to rounding only."""
    from pir_amd import scene
    _, mats, dinonet = scene.build_networks(seed=6, device="cuda")
    n, N = 5000, 9000
    pts, nrm, geo, dino = _inputs(N, seed=11)
    g = torch.Generator().manual_seed(12)
    for name in NETS:
        net = mats[name]
        w = torch.randn(N, 3 if name != "specular_roughness_network" else 1, generator=g).cuda()
        res = []
        for rows in (n, N):
            leaves = [t[:rows].clone().requires_grad_(True) for t in (pts, nrm, geo, dino)]
            net.zero_grad()
            out = _call(net, name, *leaves)
            assert out.shape[1] == w.shape[1]
            ww = w[:rows].clone()
            ww[n:] = 0
            (out * ww).sum().backward()
            res.append((out.detach()[:n], [None if l.grad is None else l.grad[:n] for l in leaves],
                        {k: p.grad.clone() for k, p in net.named_parameters()}))
        (o0, gl0, gp0), (o1, gl1, gp1) = res
        assert torch.equal(o0, o1), name
        for a, b, nm in zip(gl0, gl1, ("points", "normals", "geo_f", "dino_f")):
            assert (a is None) == (b is None), (name, nm)
            if a is not None:
                assert torch.equal(a, b), (name, nm)
        for k in gp0:
            assert rel_err(gp0[k].cpu().numpy(), gp1[k].cpu().numpy()) < 2e-5, (name, k)
    with torch.no_grad():
        assert torch.equal(dinonet(pts[:n], is_training=False), dinonet(pts, is_training=False)[:n])


def _graph_has(fn, name):
    seen, todo = set(), [fn]
    while todo:
        f = todo.pop()
        if f is None or f in seen:
            continue
        seen.add(f)
        todo.extend(n for n, _ in f.next_functions)
    return any(name in type(f).__name__ for f in seen)


@pytest.mark.parametrize("P", [300, 70001])
def test_smoothness_double_backward_matches_eager(P):
    """d material / d points differentiated once more (get_material_grads, render_surface.py:40-48; the smoothness
    loss :669-672): fused second-order backward (pir_mlp_backward_tangent) against autograd on the eager expression,
    for every parameter gradient and the gradients reaching points / features."""
    from pir_amd import scene, mlp_hip
    from pir_amd.render_fn import get_material_grads
    _, mats, _ = scene.build_networks(seed=4, device="cuda")
    pts, nrm, geo, dino = _inputs(P, seed=3)
    g = torch.Generator().manual_seed(9)
    wk = torch.rand(P, generator=g).cuda()
    for name in ("specular_albedo_network", "specular_roughness_network", "diffuse_albedo_network"):
        net = mats[name]
        res = []
        for mode in ("fused", "eager"):
            p = pts.clone().requires_grad_(True)
            gf = geo.clone().requires_grad_(True)
            net.zero_grad()
            ctx = mlp_hip.eager_grad() if mode == "eager" else torch.enable_grad()
            with ctx:
                out = _call(net, name, p, nrm, gf, dino).abs()
                if mode == "fused":
                    assert _graph_has(out.grad_fn, "_MlpFn")
                gr = get_material_grads(out, p)                       # create_graph=True
                if mode == "fused":
                    assert _graph_has(gr.grad_fn, "_MlpBwdFn")
            loss = (gr.norm(dim=-1) * wk).mean() + 0.1 * (out.sum(-1) * wk).mean()
            loss.backward()
            res.append((gr.detach(), p.grad.clone(), gf.grad.clone(), {k: q.grad.clone() for k, q in net.named_parameters()}))
        (g0, p0, f0, w0), (g1, p1, f1, w1) = res
        row = ((g0 - g1).abs().max(-1).values / g1.abs().max())
        assert (row > 1e-4).float().mean().item() < 2e-3, name          # ReLU kink flips on isolated points only
        if P <= 1000:
            assert rel_err(p0.cpu().numpy(), p1.cpu().numpy()) < 1e-3 and rel_err(f0.cpu().numpy(), f1.cpu().numpy()) < 1e-3
        errs = [rel_err(w0[k].cpu().numpy(), w1[k].cpu().numpy()) for k in w0]
        assert float(np.median(errs)) < 2e-3 and max(errs) < 2e-2, (name, errs)


def test_material_backward_large_vs_float64():
    """140 001 points (two chunks, several tiles per workgroup).  With ~10^8 ReLU units a few pre-activations land
    within rounding of zero, where the MFMA and the rocBLAS summation orders pick different sides of the kink, so
    fp32 paths are compared through a float64 evaluation of the same module: the fused kernels must be as close
    to it as the eager fp32 expression is."""
    import copy
    from pir_amd import scene, mlp_hip
    from pir_amd.network import RenderingNetwork
    P = 140001
    _, mats, _ = scene.build_networks(seed=4, device="cuda")
    for name in ("diffuse_albedo_network", "specular_albedo_network"):
        net = mats[name]
        net64 = RenderingNetwork(**scene.MODEL_CONFIG[name]).cuda()
        net64.load_state_dict(net.state_dict())
        net64 = net64.double()
        pk = mlp_hip.packed_for(net)
        torch.manual_seed(1)
        x = torch.randn(P, pk.k0, device="cuda") * 0.5
        w = torch.randn(P, pk.n_out, device="cuda")
        runs = {}
        for tag, fn, m, dt in (("fused", mlp_hip.mlp_apply_fused, net, torch.float32),
                               ("eager", mlp_hip.mlp_apply_eager, net, torch.float32),
                               ("f64", mlp_hip.mlp_apply_eager, net64, torch.float64)):
            xg = x.to(dt).detach().clone().requires_grad_(True)
            m.zero_grad()
            (fn(m, xg) * w.to(dt)).sum().backward()
            runs[tag] = (xg.grad.double(), {k: p.grad.double().clone() for k, p in m.named_parameters()})
        ref_x, ref_p = runs["f64"]
        row_err = lambda t: ((runs[t][0] - ref_x).abs().max(-1).values / ref_x.abs().max())  # noqa: E731
        assert (row_err("fused") > 1e-4).float().mean().item() < 1e-3        # isolated kink flips only ...
        assert row_err("fused").median().item() < 1e-6                       # ... every other row at rounding level
        assert (row_err("fused") > 1e-4).sum().item() <= 2 * (row_err("eager") > 1e-4).sum().item() + 10
        perr = lambda t: [((runs[t][1][k] - ref_p[k]).abs().max() / ref_p[k].abs().max()).item() for k in ref_p]  # noqa: E731
        e_f, e_e = perr("fused"), perr("eager")
        # a flipped unit moves single parameter-gradient entries by ~1e-3 of the maximum in BOTH fp32 paths
        # (measured: 6.9e-3 fused / 6.7e-3 eager worst case, 1.1e-3 / 1.0e-3 median over the parameters)
        assert max(e_f) < 2 * max(e_e) + 1e-3, (name, max(e_f), max(e_e))
        assert float(np.median(e_f)) < 2 * float(np.median(e_e)) + 2e-4, (name, np.median(e_f), np.median(e_e))


def test_fused_material_networks_match_reference_fixture():
    """Fused forward and backward at full size against the reference's own modules
    (tests/golden/materials_full.npz, generated by tools/gen_golden_materials.py)."""
    import materials_util as MU
    from conftest import load_golden
    from pir_amd import scene, mlp_hip
    from pir_amd.network import RenderingNetwork, DINONetwork
    d = load_golden("materials_full.npz")
    pts, nrm, geo, dino = (torch.from_numpy(a).cuda() for a in MU.inputs(int(d["P"]), int(d["input_seed"])))
    for name in MU.NAMES:
        net = (DINONetwork if name == "dino_network" else RenderingNetwork)(**scene.MODEL_CONFIG[name])
        net.load_state_dict(MU.fill_state(net.state_dict(), int(d[f"{name}.weight_seed"])))
        net = net.cuda()
        assert mlp_hip.module_supported(net)
        with torch.no_grad():
            y = MU.call(net, name, pts, nrm, geo, dino)
        assert rel_err(y.cpu().numpy(), d[f"{name}.out"]) < 5e-6, name
        if name == "dino_network":
            continue
        leaves = [t.clone().requires_grad_(True) for t in (pts, nrm, geo, dino)]
        y = MU.call(net, name, *leaves)
        (y * torch.from_numpy(d[f"{name}.cw"]).cuda()).sum().backward()
        for nm, t in zip(("pts", "nrm", "geo", "dino"), leaves):
            if f"{name}.g_{nm}" in d.files:
                assert rel_err(t.grad.cpu().numpy(), d[f"{name}.g_{nm}"]) < 2e-5, (name, nm)
        for k, p in net.named_parameters():
            ref = d[f"{name}.g.{k}"]
            g = p.grad.cpu().numpy()
            assert rel_err(g if g.shape == ref.shape else g[::16], ref) < 4e-5, (name, k)


def test_backward_in_small_chunks_matches_one_chunk(monkeypatch):
    """pir_mlp_backward walks the points in chunks that fit its scratch.  With scratch for 128 rows, 300 points are
    three chunks (128, 128, 44: every one a short call of 32-point tiles): same input gradients bit for bit as the
    one-chunk call, parameter gradients (accumulated chunk after chunk) to rounding."""
    from pir_amd import scene, mlp_hip, sdf_hip
    _, mats, _ = scene.build_networks(seed=8, device="cuda")
    name = "specular_albedo_network"
    net = mats[name]
    pts, nrm, geo, dino = _inputs(300, seed=13)
    w = torch.randn(300, 3, generator=torch.Generator().manual_seed(14)).cuda()
    res = []
    for chunk in (None, 128):
        mlp_hip.free_scratch()
        sdf_hip.free_scratch()
        if chunk:
            monkeypatch.setattr(mlp_hip, "BWD_CHUNK_POINTS", chunk)
        leaves = [t.clone().requires_grad_(True) for t in (pts, nrm, geo, dino)]
        net.zero_grad()
        (_call(net, name, *leaves) * w).sum().backward()
        if chunk:
            assert mlp_hip._backward_scratch(pts.device, 4, 300).numel() \
                < _scratch_bytes_for(mlp_hip, 4, 256), "the small scratch was not used"
        res.append(([l.grad for l in leaves], {k: p.grad.clone() for k, p in net.named_parameters()}))
    mlp_hip.free_scratch()
    (gl0, gp0), (gl1, gp1) = res
    for a, b, nm in zip(gl0, gl1, ("points", "normals", "geo_f", "dino_f")):
        assert (a is None) == (b is None), nm
        if a is not None:
            assert torch.equal(a, b), nm
    for k in gp0:
        assert rel_err(gp1[k].cpu().numpy(), gp0[k].cpu().numpy()) < 2e-5, k


def _scratch_bytes_for(mlp_hip, n_hidden, rows):
    from pir_amd import _lib
    return _lib.lib().pir_mlp_backward_scratch_bytes(n_hidden, rows)


def test_mlp_entry_points_reject_bad_arguments():
    """C-ABI error behaviour: malformed plans, missing pointers, scratch too small; P = 0 is accepted and does nothing."""
    import ctypes
    from pir_amd import scene, mlp_hip, _lib
    _, mats, dinonet = scene.build_networks(seed=1, device="cuda")
    pk = mlp_hip.packed_for(mats["specular_roughness_network"])
    L = _lib.lib()
    x = torch.zeros(64, 8 * pk.kin_kb, device="cuda")
    out = torch.zeros(64, pk.n_out, device="cuda")
    st = _lib.stream_ptr()
    ok = L.pir_mlp_forward(_lib.ptr(pk.buf), pk.mats_floats, ctypes.byref(pk.plan), _lib.ptr(x), 0, None, st)
    assert ok == 0                                                      # empty input
    assert L.pir_mlp_forward(_lib.ptr(pk.buf), pk.mats_floats, ctypes.byref(pk.plan), None, 64, _lib.ptr(out), st) == 1
    bad = mlp_hip.PirMlpPlan.from_buffer_copy(pk.plan)
    bad.n_hidden = 9
    assert L.pir_mlp_forward(_lib.ptr(pk.buf), pk.mats_floats, ctypes.byref(bad), _lib.ptr(x), 64, _lib.ptr(out), st) == 1
    bad = mlp_hip.PirMlpPlan.from_buffer_copy(pk.plan)
    bad.skip = 0
    assert L.pir_mlp_forward(_lib.ptr(pk.buf), pk.mats_floats, ctypes.byref(bad), _lib.ptr(x), 64, _lib.ptr(out), st) == 1
    nl = pk.n_hidden + 1
    gws = [torch.zeros(256, 8, device="cuda") for _ in range(nl)]
    wp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in gws])
    small = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    rc = L.pir_mlp_backward(_lib.ptr(pk.buf), pk.mats_floats, ctypes.byref(pk.plan), pk.k0, _lib.ptr(x), 64, _lib.ptr(out),
                            None, wp, wp, _lib.ptr(small), small.numel(), st)
    assert rc == 3                                                      # scratch too small
    # a plan without transposed products (packed for inference only) is refused by the backward
    big = mlp_hip._backward_scratch(torch.device("cuda", 0), pk.n_hidden, 64)
    bad = mlp_hip.PirMlpPlan.from_buffer_copy(pk.plan)
    bad.has_bwd = 0
    rc = L.pir_mlp_backward(_lib.ptr(pk.buf), pk.mats_floats, ctypes.byref(bad), pk.k0, _lib.ptr(x), 64, _lib.ptr(out),
                            None, wp, wp, _lib.ptr(big), big.numel(), st)
    assert rc == 1
    torch.cuda.synchronize()


@pytest.mark.parametrize("P", [65, 20000])
def test_dino_field_backward_matches_eager(P):
    """DINONetwork(is_training=True) as train_dino uses it (render_surface.py:435-492): 384 outputs through the
    tile-pass output layer, backward through the transposed output product and the shared weight-gradient GEMM."""
    from pir_amd import scene, mlp_hip
    _, _, dinonet = scene.build_networks(seed=6, device="cuda")
    pts = (torch.rand(P, 3, generator=torch.Generator().manual_seed(P)) - 0.5).cuda()
    w = torch.randn(P, 384, generator=torch.Generator().manual_seed(1)).cuda()
    res = []
    for mode in ("fused", "eager"):
        p = pts.clone().requires_grad_(True)
        dinonet.zero_grad()
        if mode == "eager":
            with mlp_hip.eager_grad():
                out = dinonet(p, is_training=True)
        else:
            out = dinonet(p, is_training=True)
            assert _graph_has(out.grad_fn, "_MlpFn")
        (out * w).sum().backward()
        res.append((out.detach(), p.grad.clone(), {k: q.grad.clone() for k, q in dinonet.named_parameters()}))
    (o0, p0, g0), (o1, p1, g1) = res
    assert rel_err(o0.cpu().numpy(), o1.cpu().numpy()) < 2e-6
    row = (p0 - p1).abs().max(-1).values / p1.abs().max()
    assert (row > 1e-4).float().mean().item() < 2e-3
    errs = [rel_err(g0[k].cpu().numpy(), g1[k].cpu().numpy()) for k in g0]
    assert float(np.median(errs)) < 5e-4 and max(errs) < 1e-2, errs


def test_blend_network_runs_zero_padded_on_the_fused_kernels():
    """BlendNetwork (128 wide, src/network.py:295-367) runs on the 256-wide engine zero-padded: forward and every
    gradient against the eager expression."""
    from pir_amd import scene, mlp_hip
    _, mats, _ = scene.build_networks(seed=2, device="cuda")
    net = mats["blend_network"]
    assert mlp_hip.module_supported(net)
    g = torch.Generator().manual_seed(4)
    P = 777
    dist, rough, cosv = (torch.rand(P, 1, generator=g).cuda() for _ in range(3))
    w = torch.randn(P, 1, generator=g).cuda()
    with torch.no_grad():
        y_inf = net(dist, rough, cosv)
    res = []
    for mode in ("fused", "eager"):
        leaves = [t.clone().requires_grad_(True) for t in (dist, rough, cosv)]
        net.zero_grad()
        if mode == "eager":
            with mlp_hip.eager_grad():
                y = net(*leaves)
        else:
            y = net(*leaves)
            assert _graph_has(y.grad_fn, "_MlpFn")
        (y * w).sum().backward()
        res.append((y.detach(), [t.grad.clone() for t in leaves], {k: p.grad.clone() for k, p in net.named_parameters()}))
    (y0, l0, g0), (y1, l1, g1) = res
    assert rel_err(y0.cpu().numpy(), y1.cpu().numpy()) < 2e-5 and rel_err(y_inf.cpu().numpy(), y1.cpu().numpy()) < 2e-5
    for a, b in zip(l0, l1):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 8e-6
    for k in g0:
        assert g0[k].shape == g1[k].shape
        assert rel_err(g0[k].cpu().numpy(), g1[k].cpu().numpy()) < 4e-5, k


def test_dino_double_backward_and_passback_stay_differentiable():
    """The 384-wide DINO field has no fused second-order path: under create_graph=True its backward must stay in the
    graph (through the eager expression) instead of silently returning constants.  passback (src/network.py:469-486)
    differentiated once more against the eager module."""
    from pir_amd import scene, mlp_hip
    _, _, dino = scene.build_networks(seed=4, device="cuda")
    g = torch.Generator().manual_seed(12)
    pts0 = (torch.rand(300, 3, generator=g) - 0.5).cuda()
    res = []
    for mode in ("fused", "eager"):
        dino.zero_grad()
        p = pts0.clone().requires_grad_(True)
        ctx = mlp_hip.eager_grad() if mode == "eager" else torch.enable_grad()
        with ctx:
            feat = dino(p, is_training=True)
            if mode == "fused":
                assert _graph_has(feat.grad_fn, "_MlpFn")
            y, grads = dino.passback(p, feat, is_training=True)
            assert grads.requires_grad                       # still differentiable
            (grads.pow(2).sum() + 1e-3 * y.pow(2).sum()).backward()
        res.append((grads.detach(), p.grad.clone(), {k: q.grad.clone() for k, q in dino.named_parameters()}))
    (g0, p0, w0), (g1, p1, w1) = res
    assert rel_err(g0.cpu().numpy(), g1.cpu().numpy()) < 1e-4
    assert rel_err(p0.cpu().numpy(), p1.cpu().numpy()) < 1e-3
    for k in w0:
        assert w0[k].abs().max().item() > 0 and rel_err(w0[k].cpu().numpy(), w1[k].cpu().numpy()) < 1e-3, k
    p = pts0.clone().requires_grad_(True)
    y, grads = dino.passback(p, dino(p, is_training=True), is_training=False)
    assert not y.requires_grad and not grads.requires_grad and grads.shape == (300, 3)
