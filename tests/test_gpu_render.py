"""GPU parity of tracer, GGX shader, SDF backward and the visibility march against the reference fixtures."""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_from_npz, rel_err
from gpu_util import sdf_net, grad_np

pytestmark = pytest.mark.gpu


# ------------------------------------------------------------------------------------------- tracer
@pytest.mark.parametrize("name,weights", [("trace_std64.npz", "sdf_full.npz"), ("trace_wide48.npz", "sdf_full.npz"),
                                          ("trace_bumpy64.npz", "sdf_full_bumpy_weights.npz"),
                                          ("trace_crop800.npz", "sdf_full_bumpy_weights.npz"),
                                          ("trace_vis32.npz", "sdf_full_bumpy_weights.npz")])
def test_trace_matches_reference(name, weights):
    from pir_amd.renderer.raytracer import RayTracer, Camera, raytrace_camera, SdfCallable, intersect_sphere
    d = load_golden(name)
    net = sdf_net(weights)
    cam = Camera(int(d["W"]), int(d["H"]), torch.from_numpy(d["K"]).cuda(), torch.from_numpy(d["W2C"]).cuda())
    tracer = RayTracer()
    res = raytrace_camera(cam, net, tracer)
    assert tracer.trace_status("cuda") == 0
    mask = res["convergent_mask"].cpu().numpy()
    ref_mask = d["cam_mask"]
    agree = (mask == ref_mask).mean()
    assert agree >= 0.9999 or (mask != ref_mask).sum() <= 1, f"hit-mask agreement {agree}"
    both = mask & ref_mask
    depth_ref = d["cam_depth"]
    err = np.abs(res["depth"].cpu().numpy() - depth_ref)[both]
    assert err.max() <= 1e-4 * depth_ref[both].max() + 1e-5, err.max()   # 1e-4 relative (north_star)
    perr = np.abs(res["points"].cpu().numpy() - d["cam_points"])[both]
    assert perr.max() < 3e-4
    # the RayTracer.forward signature on explicit rays, single chunk
    uv = cam.get_uv()
    o, dd, _ = cam.get_rays(uv)
    o, dd = o.reshape(-1, 3).contiguous(), dd.reshape(-1, 3).contiguous()
    hit, lo, hi = intersect_sphere(o, dd, 1.0)
    assert np.array_equal(hit.cpu().numpy(), d["work_mask"])
    full = tracer(SdfCallable(net), o, dd, lo, hi, hit)
    m2 = full["convergent_mask"].cpu().numpy()
    assert (m2 == d["convergent_mask"]).mean() >= 0.9999 or (m2 != d["convergent_mask"]).sum() <= 1
    ok = m2 & d["convergent_mask"]
    assert np.abs(full["distance"].cpu().numpy() - d["distance"])[ok].max() < 3e-4
    # rays the sampler gave up on come back as zeros, as in the reference
    dead = ~m2 & d["st_unfinished"] & ~d["convergent_mask"]
    if dead.any():
        assert np.abs(full["points"].cpu().numpy()[dead]).max() == 0.0


def test_trace_empty_and_all_miss():
    from pir_amd.renderer.raytracer import RayTracer, SdfCallable
    net = sdf_net("sdf_full.npz")
    tr = RayTracer()
    z = torch.zeros(0, 3, device="cuda")
    r = tr(SdfCallable(net), z, z, z[:, 0], z[:, 0], z[:, 0].bool())
    assert r["points"].shape == (0, 3)
    o = torch.tensor([[0.0, 0.0, -3.0]], device="cuda").repeat(100, 1)
    d = torch.tensor([[0.0, 1.0, 0.0]], device="cuda").repeat(100, 1)
    r = tr(SdfCallable(net), o, d, torch.zeros(100, device="cuda"), torch.zeros(100, device="cuda"),
           torch.zeros(100, dtype=torch.bool, device="cuda"))
    assert not r["convergent_mask"].any()


# ------------------------------------------------------------------------------------------- GGX
@pytest.mark.parametrize("tag,edr", [("offset", True), ("coloc", True), ("noedr", False)])
def test_ggx_matches_reference(tag, edr):
    from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer
    d = load_golden("ggx.npz")
    ggx = GGXColocatedRenderer(use_cuda=True)
    names = ("distance", "normal", "lightdir", "viewdir", "kd", "ks", "alpha")
    leaves = {n: torch.from_numpy(d[n]).cuda().requires_grad_(True) for n in names}
    light = torch.tensor(float(d["light"]), device="cuda", requires_grad=True)
    ld = leaves["viewdir"] if tag == "coloc" else leaves["lightdir"]
    o = ggx(light, leaves["distance"], leaves["normal"], ld, leaves["viewdir"], leaves["kd"], leaves["ks"],
            leaves["alpha"], EDirectRadiance=edr)
    for k in ("diffuse_rgb", "specular_rgb", "rgb"):
        ref = d[f"{tag}.{k}"]
        got = o[k].detach().cpu().numpy()
        bad = np.abs(got - ref) > 1e-4 * np.abs(ref) + 1e-4 * np.abs(ref).mean()
        worst = np.abs(got - ref)[bad].max() / np.abs(ref).mean() if bad.any() else 0.0
        print(f"ggx {tag} {k}: {int(bad.sum())} of {bad.size} elements outside 1e-4 (worst {worst:.1e} of the mean)")
        # the table bins are the reference's, point for point: no allowance for the diffuse term.  The specular term of the
        # fixture's alpha = 1e-4 edge cases has a condition number of 1 / alpha^2 = 1e8 in cos_h (D's denominator): one
        # such point (2 channels) lands 1e-3 away in the collocated case
        assert bad.sum() <= (0 if k == "diffuse_rgb" else 3), (tag, k, bad.sum())
    (o["rgb"] * torch.from_numpy(d[f"{tag}.cw"]).cuda()).sum().backward()
    assert abs(light.grad.item() - float(d[f"{tag}.g_light"])) < 2e-4 * abs(float(d[f"{tag}.g_light"]))
    for n in names:
        if tag == "coloc" and n == "lightdir":
            continue
        ref = d[f"{tag}.g_{n}"]
        got = leaves[n].grad.cpu().numpy()
        bad = np.abs(got - ref) > 2e-4 * np.abs(ref) + 1e-4 * np.abs(ref).mean() + 1e-12
        print(f"ggx {tag} d/d{n}: {int(bad.sum())} of {bad.size} elements outside 2e-4")
        assert bad.mean() < 2e-3, (tag, n, bad.sum())


def test_ggx_rgb_light_and_known_answers():
    from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer
    d = load_golden("ggx.npz")
    ggx = GGXColocatedRenderer(use_cuda=True)
    t = lambda k: torch.from_numpy(d[k]).cuda()  # noqa
    o = ggx(t("rgblight.light"), t("distance"), t("normal"), t("lightdir"), t("viewdir"), t("kd"), t("ks"), t("alpha"),
            EDirectRadiance=False)
    ref = d["rgblight.rgb"]
    bad = np.abs(o["rgb"].cpu().numpy() - ref) > 1e-4 * np.abs(ref) + 1e-4 * np.abs(ref).mean()
    assert not bad.any(), bad.sum()
    one = lambda *v: torch.tensor([list(v)], dtype=torch.float32, device="cuda")  # noqa
    nrm = one(0, 0, 1)
    o = ggx(torch.tensor(10.0), one(2.0), nrm, nrm, nrm, one(.5, .5, .5), one(.5, .5, .5), one(0.1))
    assert abs(o["diffuse_rgb"][0, 0].item() - 0.4032129943) < 1e-5
    assert abs(o["specular_rgb"][0, 0].item() - 0.3831406832) < 1e-5
    assert ggx(torch.tensor(10.0), one(2.0)[:0], nrm[:0], nrm[:0], nrm[:0], nrm[:0], nrm[:0], one(0.1)[:0])["rgb"].shape == (0, 3)


# ------------------------------------------------------------------------------------------- SDF backward
@pytest.mark.parametrize("with_feat", [True, False])
def test_sdf_get_all_backward_matches_reference(with_feat):
    d = load_golden("sdf_full.npz")
    net = sdf_net("sdf_full.npz")
    x = torch.from_numpy(d["x"]).cuda().requires_grad_(True)
    y, f, n = net.get_all(x, is_training=True)
    loss = (y * torch.from_numpy(d["cs"]).cuda()).sum() + (n * torch.from_numpy(d["cn"]).cuda()).sum()
    if with_feat:
        loss = loss + (f * torch.from_numpy(d["cf"]).cuda()).sum()
    loss.backward()
    if with_feat:
        assert abs(loss.item() - float(d["loss"])) < 1e-4 * abs(float(d["loss"])) + 1e-4
    gx_ref = d["grad_x"] if with_feat else d["grad_x_nofeat"]
    assert rel_err(x.grad.cpu().numpy(), gx_ref) < 2e-5
    pre = "g." if with_feat else "g2."
    for k, p in net.named_parameters():
        ref = d[pre + k]
        assert rel_err(grad_np(p, ref.shape), ref) < 4e-5, k


def test_sdf_backward_chunking_is_consistent():
    """Two chunk sizes (one forcing several passes and a ragged tail) give the same gradients."""
    from pir_amd import sdf_hip
    net = sdf_net("sdf_full_bumpy_weights.npz")
    g = torch.Generator().manual_seed(3)
    M = 5000
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    gs, gn = torch.randn(M, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
    a = sdf_hip.get_all_backward(net.packed(), x, gs, None, gn, True, chunk_points=131072)
    b = sdf_hip.get_all_backward(net.packed(), x, gs, None, gn, True, chunk_points=2048)
    assert torch.equal(a[2], b[2])
    for wa, wb in zip(a[0] + a[1], b[0] + b[1]):
        assert rel_err(wb.cpu().numpy(), wa.cpu().numpy()) < 2e-5


def test_sdf_backward_chunking_with_feature_gradients():
    """Same as above with an upstream gradient on the 256 features: the caller-owned g_feat rows of every chunk
    (ragged last chunk included) feed the lin8 weight-gradient jobs directly."""
    from pir_amd import sdf_hip
    net = sdf_net("sdf_full_bumpy_weights.npz")
    g = torch.Generator().manual_seed(5)
    M = 4999
    x = (torch.rand(M, 3, generator=g) * 2 - 1).cuda()
    gs, gn = torch.randn(M, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
    gf = torch.randn(M, 256, generator=g).cuda()
    a = sdf_hip.get_all_backward(net.packed(), x, gs, gf, gn, True, chunk_points=131072)
    b = sdf_hip.get_all_backward(net.packed(), x, gs, gf, gn, True, chunk_points=2048)
    assert torch.equal(a[2], b[2])
    for wa, wb in zip(a[0] + a[1], b[0] + b[1]):
        assert rel_err(wb.cpu().numpy(), wa.cpu().numpy()) < 2e-5


# ------------------------------------------------------------------------------------------- visibility
@pytest.mark.parametrize("n_samples", [128, 256])
def test_light_visibility_matches_reference(n_samples):
    from pir_amd.renderer.raytracer import Camera, light_visibility
    d = load_golden("visibility.npz")
    net = sdf_net("sdf_full_bumpy_weights.npz")
    cam = Camera(32, 32, torch.from_numpy(d["K"]).cuda(), torch.from_numpy(d["W2C"]).cuda())
    off = torch.tensor([[0.2, 0.1, 0.0]], device="cuda", requires_grad=True)
    lp = cam.get_light_pos(off)
    pts = torch.from_numpy(d["points"]).cuda()
    inv_s = torch.from_numpy(d["inv_s"]).cuda()
    vis = light_visibility(lp, pts, net, inv_s, n_samples=n_samples, is_training=True)
    assert rel_err(vis.detach().cpu().numpy(), d[f"vis{n_samples}"]) < 1e-5
    (vis * torch.from_numpy(d[f"c{n_samples}"]).cuda()).sum().backward()
    assert rel_err(off.grad.cpu().numpy(), d[f"grad_offset{n_samples}"]) < 4e-5
    for k, p in net.named_parameters():
        ref = d[f"g{n_samples}." + k]
        assert rel_err(grad_np(p, ref.shape), ref) < 4e-5, k
    v2 = light_visibility(torch.from_numpy(d["light_pos_side"]).cuda(), pts, net, inv_s, 128, is_training=False)
    if n_samples == 128:
        assert rel_err(v2.cpu().numpy(), d["vis_side128"]) < 1e-5 and not v2.requires_grad


def test_visibility_kernels_match_torch_formula_and_edge_cases():
    """pir_visibility_* against the same march written with PyTorch ops on 5000 points (both use the HIP SDF
    kernels), forward and every gradient; saturated samples must get exactly-zero gradients (the SDF backward
    skips them); empty input, the 512-sample maximum and its rejection beyond."""
    import ctypes
    from pir_amd import _lib, visibility_hip
    from pir_amd.renderer import raytracer as rt
    net = sdf_net("sdf_full_bumpy_weights.npz")
    g = torch.Generator().manual_seed(11)
    pts = (torch.nn.functional.normalize(torch.randn(5000, 3, generator=g), dim=-1) *
           (0.45 + 0.1 * torch.rand(5000, 1, generator=g))).cuda()
    inv_s = torch.tensor([[20.0855]], device="cuda")
    w = torch.rand(5000, generator=g).cuda()
    outs = []
    for fn in (rt.light_visibility, rt._light_visibility_torch):
        net.zero_grad()
        lp = torch.tensor([[0.3, 0.2, -2.9]], device="cuda", requires_grad=True)
        vis = fn(lp, pts, net, inv_s, 128, True)
        (vis * w).sum().backward()
        outs.append((vis.detach(), lp.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}))
    (v0, l0, g0), (v1, l1, g1) = outs
    assert (v0 - v1).abs().max().item() < 2e-5
    assert 0.02 < (v0 < 0.5).float().mean().item() < 0.98          # the case has lit and shadowed points
    assert rel_err(l0.cpu().numpy(), l1.cpu().numpy()) < 2e-5
    for k in g0:
        assert rel_err(g0[k].cpu().numpy(), g1[k].cpu().numpy()) < 2e-5, k

    # exact zeros where both sigmoids saturate
    bs, n = 64, 118
    sdf = torch.full((bs * n,), 2.0, device="cuda")
    nrm = torch.randn(bs * n, 3, device="cuda")
    d = torch.nn.functional.normalize(torch.randn(bs, 3, device="cuda"), dim=-1)
    L = torch.full((bs, 1), 2.5, device="cuda")
    s_, n_ = sdf.clone().requires_grad_(True), nrm.clone().requires_grad_(True)
    vis = visibility_hip.composite(s_, n_, d, L, inv_s, 128)
    vis.sum().backward()
    assert torch.count_nonzero(s_.grad).item() == 0 and torch.count_nonzero(n_.grad).item() == 0
    assert (vis - 1.0).abs().max().item() < 2e-3

    # empty input, maximum march length (n = 512), rejection beyond it
    assert rt.light_visibility(torch.zeros(1, 3, device="cuda"), torch.zeros(0, 3, device="cuda"), net, inv_s,
                               128, False).shape == (0,)
    p8 = pts[:8]
    lp = torch.tensor([[0.3, 0.2, -2.9]], device="cuda")
    big = rt.light_visibility(lp, p8, net, inv_s, 522, False)
    ref = rt._light_visibility_torch(lp, p8, net, inv_s, 522, False)
    assert (big - ref).abs().max().item() < 5e-5
    Lib = _lib.lib()
    rc = Lib.pir_visibility_forward(_lib.ptr(sdf), _lib.ptr(nrm), _lib.ptr(d), _lib.ptr(L.reshape(-1).contiguous()),
                                    _lib.ptr(inv_s.reshape(1)), 1, 523, _lib.ptr(vis.detach()), _lib.stream_ptr())
    assert rc == 1


def test_two_streams_overlap_without_sharing_scratch():
    """Two get_all + get_all_backward calls enqueued on two streams of one GPU (they overlap: each is long enough to
    still run when the other is launched) give the results of the serial calls bit for bit: scratch and workspace
    caches are keyed by (device, stream)."""
    from pir_amd import sdf_hip
    net = sdf_net("sdf_full_bumpy_weights.npz")
    pk = net.packed()
    g = torch.Generator().manual_seed(21)
    M = 150_000
    xs = [(torch.rand(M, 3, generator=g) * 2 - 1).cuda() for _ in range(2)]
    gs = [torch.randn(M, generator=g).cuda() for _ in range(2)]
    gn = [torch.randn(M, 3, generator=g).cuda() for _ in range(2)]
    serial = []
    for i in range(2):
        fw = pk.get_all(xs[i], want_feature=False)
        serial.append((fw, sdf_hip.get_all_backward(pk, xs[i], gs[i], None, gn[i], True)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    par = [None, None]
    for rep in range(2):                      # second round: the per-stream buffers already exist
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                fw = pk.get_all(xs[i], want_feature=False)
                par[i] = (fw, sdf_hip.get_all_backward(pk, xs[i], gs[i], None, gn[i], True))
        torch.cuda.synchronize()
        for i in range(2):
            (s0, _, n0), (w0, b0, x0) = serial[i]
            (s1, _, n1), (w1, b1, x1) = par[i]
            assert torch.equal(s0, s1) and torch.equal(n0, n1) and torch.equal(x0, x1), (rep, i)
            for a, b in zip(w0 + b0, w1 + b1):
                assert torch.equal(a, b), (rep, i)
    keys = {k[:2] for k in sdf_hip._bwd_scratch_cache}
    assert len({k[1] for k in keys}) >= 3      # default stream + the two side streams each own a buffer
    # the per-stream buffers are sized to the calls made on them: nowhere near the 8.7 GB maximum
    side = [b.numel() for k, b in sdf_hip._bwd_scratch_cache.items() if k[1] in {int(st.cuda_stream) for st in streams}]
    assert side and max(side) <= sdf_hip._lib.lib().pir_sdf_backward_scratch_bytes(
        max(4096, 1 << (int(xs[0].shape[0]) - 1).bit_length()))
    sdf_hip.free_scratch()
    assert not sdf_hip._bwd_scratch_cache and not sdf_hip._scratch_cache


def test_ggx_table_bins_equal_the_reference_bit_for_bit():
    """Integer work must be bit-exact: the table bins pir_ggx_forward / _backward use (floor(cos^0.25 * 100),
    floor((alpha/4)^0.25 * 50), clamped) against the reference's own torch.floor outputs on the 4096 points of ggx.npz
    (indices.npz; clamp-edge cosines, alpha below / above the table range included), both light configurations."""
    from pir_amd import _lib
    d, ix = load_golden("ggx.npz"), load_golden("indices.npz")
    P = d["normal"].shape[0]
    dev = {k: torch.from_numpy(d[k]).cuda().contiguous() for k in ("normal", "lightdir", "viewdir", "alpha")}
    total = 0
    for tag, ldir in (("offset", "lightdir"), ("coloc", "viewdir")):
        out = [torch.empty(P, dtype=torch.int32, device="cuda") for _ in range(3)]
        rc = _lib.lib().pir_ggx_table_indices(P, _lib.ptr(dev["normal"]), _lib.ptr(dev[ldir]), _lib.ptr(dev["viewdir"]),
                                              _lib.ptr(dev["alpha"]), 5000, 50, *[_lib.ptr(t) for t in out],
                                              _lib.stream_ptr())
        assert rc == 0
        for t, key in zip(out, ("idx_wi", "idx_wo", "idx_diff")):
            ref = ix[f"ggx.{tag}.{key}"]
            bad = np.nonzero(t.cpu().numpy() != ref)[0]
            total += bad.size
            assert bad.size == 0, (tag, key, bad.size, bad[:8], t.cpu().numpy()[bad[:8]], ref[bad[:8]])
    print("GGX table bins: 0 of", 6 * P, "differ from the reference's")


@pytest.mark.parametrize("iters", [0, 16])
@pytest.mark.parametrize("name", ["trace_crop800", "trace_bumpy64", "trace_std64"])
def test_sampler_first_negative_index_equals_the_reference_min_idx(name, iters):
    """The dense sampler's index work (raytracer.py:192-197): the first negative sample found by the HIP sampler
    against the reference's min_idx (where its min_val < 0; rays without a negative sample must report none).
    iters = 0: no sphere tracing, so every ray enters the sampler with the fixture's own interval [min_dis, max_dis]:
    identical inputs, the index must be EQUAL except where the disputed sample's |sdf| is round-off (< 1e-5: its sign
    is undecidable in fp32).  iters = 16 (the default tracer): the same comparison on the rays both tracers hand to the
    sampler; on the 16x zoomed silhouette crop, grazing rays oscillate around the surface for 16 signed steps and their
    accumulated distance -- the sampler's interval -- differs between any two fp32 evaluations: the reference run on one
    CPU thread, or with the network called on slices of 997 rows, disagrees with its own 8-thread run on 44 of 1 848
    rays there (0 on the two 64 x 64 views).  The bound is twice that self-disagreement (so 0 where it is 0)."""
    from pir_amd import _lib
    from pir_amd.renderer import raytracer as rt
    d, ix = load_golden(name + ".npz"), load_golden("indices.npz")
    net = sdf_net("sdf_full.npz" if name == "trace_std64" else "sdf_full_bumpy_weights.npz")
    cam = rt.Camera(int(d["W"]), int(d["H"]), torch.from_numpy(d["K"]).cuda(), torch.from_numpy(d["W2C"]).cuda())
    ray_o, ray_d, _ = cam.get_rays(cam.get_uv())
    ray_o, ray_d = ray_o.reshape(-1, 3).contiguous(), ray_d.reshape(-1, 3).contiguous()
    n = ray_o.shape[0]
    tracer = rt.RayTracer(sphere_tracing_iters=iters)
    res = tracer(rt.SdfCallable(net), ray_o, ray_d, torch.from_numpy(d["min_dis"]).cuda(),
                 torch.from_numpy(d["max_dis"]).cuda(), torch.from_numpy(d["work_mask"]).cuda())
    key = name + (".it0" if iters == 0 else "")
    ref_conv = ix[key + ".convergent_mask"] if iters == 0 else d["convergent_mask"]
    assert (res["convergent_mask"].cpu().numpy() != ref_conv).sum() <= (0 if iters else 2)
    ws = rt._trace_ws[_lib.stream_key(ray_o.device)]
    idx = torch.empty(n, dtype=torch.int32, device="cuda")
    assert _lib.lib().pir_trace_sampler_index(_lib.ptr(ws), n, 0, tracer.max_bisect, _lib.ptr(idx), _lib.stream_ptr()) == 0
    idx = idx.cpu().numpy()
    rays, ref_idx, neg = ix[key + ".sampler_rays"], ix[key + ".min_idx"], ix[key + ".min_val_negative"]
    in_sampler = np.zeros(n, bool)
    in_sampler[rays] = True
    # which rays reach the sampler is a float decision (|sdf| > 5e-5), not index work
    other_side = int(((idx != -2) != in_sampler).sum())
    assert other_side <= max(2, rays.size // 500), other_side
    common = idx[rays] != -2
    got = idx[rays]
    ref_first = np.where(neg, ref_idx, -1)                  # the reference's first negative sample, -1 = none
    differ = np.nonzero(common & (got != ref_first))[0]
    worst = 0.0
    if differ.size and iters == 0:
        # identical sampler inputs: the earlier of the two indices is the disputed sample, its |sdf| must be round-off
        r = rays[differ]
        lo, hi = d["min_dis"].reshape(-1)[r].astype(np.float32), d["max_dis"].reshape(-1)[r].astype(np.float32)
        a, b = got[differ].astype(np.int64), ref_first[differ].astype(np.int64)
        k = np.where((a >= 0) & ((b < 0) | (a < b)), a, b)
        z = lo + torch.linspace(0, 1, tracer.n_steps).numpy()[k] * (hi - lo)
        sel = torch.from_numpy(r).cuda().long()
        with torch.no_grad():
            v = net.sdf(ray_o[sel] + ray_d[sel] * torch.from_numpy(z).cuda()[:, None])[:, 0].abs().cpu().numpy()
        worst = float(v.max())
        assert worst < 1e-5, (worst, differ.size)
    own = 0
    if iters == 0:
        assert differ.size <= max(2, rays.size // 200), differ.size
    else:
        # yardstick: the reference against ITSELF under two other fp32 evaluation orders (1 CPU thread; the network
        # called on slices of 997 rows), same rays, same default tracer (tools/gen_golden.py, indices section)
        ref_map = dict(zip(rays.tolist(), ref_first.tolist()))
        for alt in ("alt_t1", "alt_rows997"):
            r2 = ix[f"{key}.{alt}.sampler_rays"]
            f2 = np.where(ix[f"{key}.{alt}.min_val_negative"], ix[f"{key}.{alt}.min_idx"], -1)
            own = max(own, sum(ref_map[r] != f for r, f in zip(r2.tolist(), f2.tolist()) if r in ref_map))
        assert differ.size <= 2 * own, (differ.size, own)     # 0 allowed only where the reference agrees with itself
    print(f"{key}: {rays.size} rays in the reference's sampler ({other_side} on the other side of the threshold here); "
          f"first-negative index equal on {int(common.sum()) - differ.size} of {int(common.sum())} common rays"
          + (f"; disputed samples have |sdf| <= {worst:.1e}" if differ.size and iters == 0 else "")
          + (f"; the reference differs from itself on {own} rays when its fp32 evaluation order changes" if iters else ""))


def test_ggx_properties_at_four_million_points():
    """Size-independent properties of the shader at 4 M + 3 points (register kernel + scalar tail): exact homogeneity
    (light x 2 doubles every output bit for bit; kd x 0.5 halves the diffuse term and leaves the specular term
    untouched), the same values from unaligned buffers (LDS-staged kernel) as from aligned ones (register kernel), the
    tail points equal to the same points evaluated alone, and rgb = diffuse + specular exactly."""
    from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer
    ggx = GGXColocatedRenderer(use_cuda=True)
    P = (1 << 22) + 3
    g = torch.Generator(device="cuda").manual_seed(9)
    unit = lambda t: torch.nn.functional.normalize(t, dim=-1)  # noqa: E731
    n = unit(torch.randn(P, 3, device="cuda", generator=g))
    l = unit(n + 0.7 * torch.randn(P, 3, device="cuda", generator=g))
    v = unit(n + 0.7 * torch.randn(P, 3, device="cuda", generator=g))
    dist = 2.0 + torch.rand(P, 1, device="cuda", generator=g)
    kd, ks = torch.rand(P, 3, device="cuda", generator=g), torch.rand(P, 3, device="cuda", generator=g)
    alpha = 0.01 + 0.6 * torch.rand(P, 1, device="cuda", generator=g)
    light = torch.tensor(135.0, device="cuda")
    with torch.no_grad():
        o = ggx(light, dist, n, l, v, kd, ks, alpha)
        o2 = ggx(light * 2, dist, n, l, v, kd, ks, alpha)
        o3 = ggx(light, dist, n, l, v, kd * 0.5, ks, alpha)
        assert all(torch.equal(o2[k], 2 * o[k]) for k in o)
        assert torch.equal(o3["diffuse_rgb"], 0.5 * o["diffuse_rgb"]) and torch.equal(o3["specular_rgb"], o["specular_rgb"])
        assert rel_err(o["rgb"].cpu(), (o["diffuse_rgb"] + o["specular_rgb"]).cpu()) < 2e-7       # one rounding (the sum may be fused)
        assert torch.isfinite(o["rgb"]).all() and (o["diffuse_rgb"] >= 0).all() and (o["specular_rgb"] >= 0).all()

        def shifted(t):      # the same values at an address that is NOT 16-byte aligned
            flat = torch.empty(t.numel() + 1, device="cuda")
            flat[1:] = t.reshape(-1)
            return flat[1:].view(t.shape)
        # straight through the C ABI (the Python mirror would copy an unaligned tensor to an aligned one)
        from pir_amd import _lib
        t_rough, t_diff = ggx._tables_on(n.device)
        ns = shifted(n)
        assert ns.data_ptr() % 16 != 0
        outs = [torch.empty(P, 3, device="cuda") for _ in range(3)]
        args = [dist.reshape(P), ns, l, v, kd, ks, alpha.reshape(P)]
        assert _lib.lib().pir_ggx_forward(P, _lib.ptr(light.reshape(1)), 0, 1, *[_lib.ptr(a) for a in args],
                                          _lib.ptr(t_rough), t_rough.numel(), _lib.ptr(t_diff), t_diff.numel(),
                                          *[_lib.ptr(t) for t in outs], _lib.stream_ptr()) == 0
        for k, got in zip(("diffuse_rgb", "specular_rgb", "rgb"), outs):
            # two kernels, same source arithmetic: the compiler contracts the cosines differently (1 ulp), which moves the
            # few points that sit within an ulp of a table-bin edge into the neighbouring bin (a 1e-3 step of the diffuse term)
            off = ((got - o[k]).abs() > 1e-6 * o[k].abs().max()).sum().item()
            print(f"ggx unaligned-buffer kernel vs register kernel, {k}: {off} of {got.numel()} elements differ by > 1e-6")
            assert off < 1e-4 * got.numel(), (k, off)
            assert rel_err(got.cpu(), o[k].cpu()) < 1e-3, k
        tail = slice(P - 3, P)
        ot = ggx(light, dist[tail], n[tail], l[tail], v[tail], kd[tail], ks[tail], alpha[tail])
        assert all(torch.equal(ot[k], o[k][tail]) for k in o)
