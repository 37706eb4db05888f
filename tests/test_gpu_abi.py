"""Direct C-ABI checks of the small entry points against the reference fixtures and of
the tracer's bisection bound (row a7: the reference loops until every bracket is below 2*threshold)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_util import sdf_net

pytestmark = pytest.mark.gpu


def _elem_rel(got, ref, floor):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return float((np.abs(got - ref) / (np.abs(ref) + floor)).max())


@pytest.mark.parametrize("mode", ["uv", "pixel_grid"])
def test_camera_rays_abi_matches_reference(mode):
    """pir_camera_rays vs Camera.get_rays + intersect_sphere captured from the reference (camera.npz)."""
    from pir_amd import _lib
    d = load_golden("camera.npz")
    W, H = int(d["W"]), int(d["H"])
    n = W * H
    from pir_amd.renderer.raytracer import Camera
    # K_inv / C2W as the reference forms them: fp32 torch.inverse on the host (raytracer.py:277-289)
    cam21 = Camera(W, H, torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"])).cam21().cuda()
    uv = torch.from_numpy(d["uv"]).reshape(-1, 2).contiguous().cuda()
    o, dd = torch.empty(n, 3, device="cuda"), torch.empty(n, 3, device="cuda")
    nrm, lo, hi = torch.empty(n, device="cuda"), torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    hit = torch.empty(n, dtype=torch.uint8, device="cuda")
    P = _lib.ptr
    rc = _lib.lib().pir_camera_rays(n, W, P(uv) if mode == "uv" else None, P(cam21), 1.0, P(o), P(dd), P(nrm), P(hit),
                                    P(lo), P(hi), _lib.stream_ptr())
    assert rc == 0
    torch.cuda.synchronize()
    errs = {"ray_o": np.abs(o.cpu().numpy() - d["ray_o"].reshape(-1, 3)).max(),
            "ray_d": np.abs(dd.cpu().numpy() - d["ray_d"].reshape(-1, 3)).max(),      # unit vectors: absolute
            "ray_d_norm": _elem_rel(nrm.cpu().numpy(), d["ray_d_norm"].reshape(-1), 1e-12)}
    m = d["mask"].astype(bool)
    assert np.array_equal(hit.cpu().numpy().astype(bool), m)
    errs["min_dis"] = _elem_rel(lo.cpu().numpy()[m], d["min_dis"][m], 1e-3)
    errs["max_dis"] = _elem_rel(hi.cpu().numpy()[m], d["max_dis"][m], 1e-3)
    print("camera_rays", mode, errs)
    assert errs["ray_o"] == 0.0
    assert errs["ray_d"] < 2e-7 and errs["ray_d_norm"] < 2e-7
    # the sphere interval is a difference of nearly equal numbers for grazing rays: 1e-5 element-relative
    assert errs["min_dis"] < 1e-5 and errs["max_dis"] < 1e-5


def test_reparam_points_gpu_matches_reference():
    """reparam_points value and gradient on the GPU vs reparam.npz (raytracer.py:18-25)."""
    from pir_amd.renderer.raytracer import reparam_points
    d = load_golden("reparam.npz")
    t = lambda k: torch.from_numpy(d[k]).cuda()  # noqa: E731
    sdf = t("sdf").requires_grad_(True)
    out = reparam_points(t("points"), t("grads"), t("dirs"), sdf)
    assert torch.equal(out.detach().cpu(), torch.from_numpy(d["out"]))           # value-identical to the points
    (out * t("cw")).sum().backward()
    e = _elem_rel(sdf.grad.cpu().numpy(), d["g_sdf"], 1e-6 * np.abs(d["g_sdf"]).max())
    print("reparam grad element-relative", e)
    assert e < 2e-6


def test_bisection_runs_as_many_rounds_as_the_reference():
    """A 381-long sampler interval starts bisection from a 3.0-wide bracket: 15 halvings to reach 2*5e-5 (the old
    fixed bound of 12 would have stopped at 7e-4).  The HIP tracer must agree with the loop-until-done reference
    logic (RayTracer._trace_torch on the same SDF kernels) and report a clean status; with the bound forced down to
    12 rounds the status check must raise."""
    from pir_amd.renderer.raytracer import RayTracer, SdfCallable
    net = sdf_net("sdf_full.npz")
    n = 300
    g = torch.Generator().manual_seed(2)
    o = torch.tensor([[0.0, 0.0, -3.0]]).repeat(n, 1)
    o[:, :2] += 0.2 * (torch.rand(n, 2, generator=g) - 0.5)
    o, d = o.cuda(), torch.tensor([[0.0, 0.0, 1.0]], device="cuda").repeat(n, 1)
    lo, hi = torch.zeros(n, device="cuda"), torch.full((n,), 381.0, device="cuda")
    work = torch.ones(n, dtype=torch.bool, device="cuda")
    tr = RayTracer(sphere_tracing_iters=0)           # every ray goes to the sampler
    assert tr.max_bisect >= 15
    ref = tr._trace_torch(SdfCallable(net), o, d, lo.clone(), hi.clone(), work)
    got = tr(SdfCallable(net), o, d, lo, hi, work)
    tr.check_status()                                 # clean
    assert bool(ref["convergent_mask"].all()) and torch.equal(got["convergent_mask"], ref["convergent_mask"])
    err = (got["distance"] - ref["distance"]).abs().max().item()
    print("bisection distance error", err, "sdf at the hit", got["sdf"].abs().max().item())
    assert err <= 1e-4 and got["sdf"].abs().max().item() < 2e-4
    tr.max_bisect = 12
    tr(SdfCallable(net), o, d, lo, hi, work)
    with pytest.raises(RuntimeError, match="bisection bound"):
        tr.check_status()
    tr.check_status()                                 # the flag was reset by the failed check


@pytest.mark.parametrize("collocated", [False, True])
def test_shade_inputs_kernel_matches_torch_expressions(collocated):
    """pir_shade_inputs_* (reparam_points + normal normalisation + lightdir / distance fused) against the reference's
    eager expressions (raytracer.py:18-25, :830-833, render_surface.py:121, :134-135): values bit-exact, gradients
    to sdf, normal and the light position element-wise."""
    import torch.nn.functional as F
    from pir_amd import shade_hip
    from pir_amd.renderer.raytracer import reparam_points
    g = torch.Generator().manual_seed(13)
    P = 10_007
    pts = (F.normalize(torch.randn(P, 3, generator=g), dim=-1) * 0.5).cuda()
    ray_o = torch.tensor([[0.0, 0.0, -3.0]], device="cuda").repeat(P, 1)
    ray_d = F.normalize(pts - ray_o, dim=-1)
    nrm0 = (pts * 2.1 + 0.2 * torch.randn(P, 3, generator=g).cuda())
    nrm0[:50] = -nrm0[:50]                       # back-facing: the 1e-4 clamp of reparam_points is active
    sdf0 = 1e-5 * torch.randn(P, 1, generator=g).cuda()
    w = [torch.randn(P, 3, generator=g).cuda() for _ in range(3)] + [torch.randn(P, 1, generator=g).cuda()]
    outs = []
    for fused in (True, False):
        sdf, nrm = sdf0.clone().requires_grad_(True), nrm0.clone().requires_grad_(True)
        lp = torch.tensor([[0.2, 0.1, -3.0]], device="cuda", requires_grad=True)
        if fused:
            p, nh, ld, dist = shade_hip.shade_inputs(pts, sdf, nrm, ray_d, ray_o, None if collocated else lp, True)
        else:
            p = reparam_points(pts, nrm.detach(), -ray_d, sdf)
            nh = nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-10)
            org = ray_o if collocated else lp
            ld = None if collocated else F.normalize(p - org, eps=1e-10)
            dist = (p - org).norm(dim=-1, keepdim=True)
        loss = (p * w[0]).sum() + (nh * w[1]).sum() + (dist * w[3]).sum()
        if ld is not None:
            loss = loss + (ld * w[2]).sum()
        loss.backward()
        outs.append((p.detach(), nh.detach(), None if ld is None else ld.detach(), dist.detach(), sdf.grad, nrm.grad,
                     lp.grad))
    a, b = outs
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], pts)
    for i, name in ((1, "nhat"), (2, "lightdir"), (3, "distance")):
        if a[i] is not None:
            rel = ((a[i] - b[i]).abs() / b[i].abs().clamp(min=1.0)).max().item()
            assert rel <= 1.2e-7, (name, rel)                              # 1 ulp (unit vectors; |v| ~ 3)
    for i, name in ((4, "g_sdf"), (5, "g_normal"), (6, "g_light_pos")):
        if a[i] is None or b[i] is None:
            assert a[i] is None and b[i] is None or collocated, name
            continue
        err = ((a[i] - b[i]).abs() / (b[i].abs() + b[i].abs().mean())).max().item()
        print("shade_inputs", "collocated" if collocated else "offset", name, err)
        assert err < 2e-5, (name, err)


def test_shade_inputs_abi_edge_cases():
    """P = 0 does nothing, missing required pointers are rejected, and render_normal_and_color on a view without a
    single hit (fused path never entered) returns all-zero images."""
    from pir_amd import _lib, scene
    from pir_amd.render_fn import make_render_fn
    from pir_amd.renderer.raytracer import RayTracer, Camera, render_camera
    from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer
    L = _lib.lib()
    S = _lib.stream_ptr()
    assert L.pir_shade_inputs_forward(0, None, None, None, 0, None, None, None, S) == 0
    assert L.pir_shade_inputs_backward(0, None, None, None, None, 0, None, None, None, None, None, None, None, S) == 0
    t = torch.zeros(4, 3, device="cuda")
    assert L.pir_shade_inputs_forward(4, _lib.ptr(t), None, _lib.ptr(t), 0, _lib.ptr(t), None, _lib.ptr(t), S) == 1
    assert L.pir_shade_inputs_forward(-1, _lib.ptr(t), _lib.ptr(t), _lib.ptr(t), 0, _lib.ptr(t), None, _lib.ptr(t), S) == 1
    # g_sdf wanted but no ray_d given
    assert L.pir_shade_inputs_backward(4, _lib.ptr(t), _lib.ptr(t), None, _lib.ptr(t), 0, None, None, None, None,
                                       _lib.ptr(t), _lib.ptr(t[:, 0].contiguous()), None, S) == 1
    sdf, mats, dino = scene.build_networks(seed=0, device="cuda")
    cam0 = scene.make_camera(16, 16, device="cuda")
    W2C = cam0.W2C.clone()
    W2C[0, 3] = 50.0                                   # the object is far outside the frustum: no ray hits the unit sphere
    cam = Camera(16, 16, cam0.K, W2C)
    res = render_camera(cam, sdf, RayTracer(), mats, dino, make_render_fn(GGXColocatedRenderer(use_cuda=True)),
                        fill_holes=False, handle_edges=False, inv_s=torch.full((1, 1), scene.INV_S, device="cuda"),
                        render_visibility=True, light_pos=cam.get_light_pos(mats["point_light_network"].offset),
                        indirect_warmup=True, is_training=True)
    assert not res["convergent_mask"].any()
    assert res["color"].abs().max().item() == 0 and res["vis"].abs().max().item() == 0 and res["normal"].shape == (16, 16, 3)


@pytest.mark.parametrize("D,L", [(3, 10), (3, 6), (3, 4), (1, 6), (3, 0)])
def test_embed_kernels_match_the_eager_encoding_to_second_order(D, L):
    """pir_embed_forward / _backward / _tangent / _second against the reference's eager expression
    (src/models/embedder.py:20-41) under autograd: values bit-identical (same sinf / cosf), first derivatives and the
    derivative of the first derivative (the smoothness loss's create_graph path) within 1e-6."""
    from pir_amd import _lib
    from pir_amd.models import embedder
    enc = embedder.FrequencyEncoding(L, D)
    g = torch.Generator().manual_seed(100 * D + L)
    x0 = ((torch.rand(1537, D, generator=g) * 2 - 1) * 1.3).cuda()
    c0 = torch.randn(1537, enc.out_dim, generator=g).cuda()
    c2 = torch.randn(1537, enc.out_dim, generator=g).cuda()
    w = torch.randn(1537, D, generator=g).cuda()
    res = []
    for fused in (True, False):
        embedder.FUSED = fused
        try:
            x = x0.clone().requires_grad_(True)
            c = c0.clone().requires_grad_(True)
            out = enc(x)
            (gx,) = torch.autograd.grad((out * c).sum(), x, create_graph=True)
            ((gx * w).sum() + (out * c2).sum()).backward()
            res.append((out.detach(), gx.detach(), x.grad.clone(), c.grad.clone()))
        finally:
            embedder.FUSED = True
    (o1, g1, xg1, cg1), (o0, g0, xg0, cg0) = res
    assert torch.equal(o1, o0)
    for a, b, name in ((g1, g0, "first derivative"), (xg1, xg0, "d/dx of it"), (cg1, cg0, "d/dg of it")):
        err = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()
        assert err < 1e-6, (name, err)
    # 3-D leading shape and a non-contiguous view go through the same kernels
    x3 = x0[:1536].reshape(4, 384, D)
    assert torch.equal(enc(x3), o1[:1536].reshape(4, 384, -1))
    assert enc(x0[:0]).shape == (0, enc.out_dim)
    # raw C ABI: zero points is a no-op, bad sizes are refused
    L_ = _lib.lib()
    assert L_.pir_embed_forward(0, D, L, None, None, None) == 0
    assert L_.pir_embed_forward(4, 0, L, _lib.ptr(x0), _lib.ptr(o1), None) == 1
    assert L_.pir_embed_backward(4, D, 17, _lib.ptr(x0), _lib.ptr(c0), _lib.ptr(w), None) == 1


def test_abi_error_conventions_zero_sizes_and_null_pointers():
    """SURVEY 8(b), error conventions of the C ABI on the GPU box: sizes of 0 are accepted and do nothing, a negative
    size or a missing required pointer is PIR_ERR_ARG, returned BEFORE anything is launched (tests/abi_util.py; the same
    cases run without a GPU in tests/test_host_logic.py)."""
    import abi_util
    from pir_amd import _lib
    buf = torch.zeros(4096, device="cuda")
    abi_util.check_error_conventions(_lib.lib(), buf.data_ptr(), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert float(buf.abs().sum()) == 0.0                                  # nothing was written
