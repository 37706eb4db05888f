"""pir_trace_static (the allocation-free tracer entry point) against pir_trace, through the C ABI.

Collected last (file name) so that a failure here cannot hide the other GPU tests under `pytest -x`."""
import numpy as np
import pytest
import torch

from gpu_util import sdf_net

pytestmark = pytest.mark.gpu


def _rays(res, focal_mul=2.0):
    from pir_amd import scene
    from pir_amd.renderer.raytracer import _camera_rays_hip
    cam = scene.make_camera(res, res, focal_mul=focal_mul, device="cuda")
    o, d, _, hit, lo, hi = _camera_rays_hip(cam, cam.get_uv(), True)
    return o, d, lo, hi, hit.to(torch.uint8)


def _outputs(n):
    return (torch.full((n,), 7, dtype=torch.uint8, device="cuda"), torch.full((n, 3), 7.0, device="cuda"),
            torch.full((n,), 7.0, device="cuda"), torch.full((n,), 7.0, device="cuda"))


def _trace_both(net, rays, chunk=0, max_bisect=12):
    from pir_amd import _lib
    L = _lib.lib()
    o, d, lo, hi, wm = rays
    n = lo.shape[0]
    pk = net.packed().buf
    ws = torch.empty(L.pir_trace_workspace_bytes(n, chunk, max_bisect), dtype=torch.uint8, device="cuda")
    a, b = _outputs(n), _outputs(n)
    status = torch.full((1,), -1, dtype=torch.int32, device="cuda")
    P, S = _lib.ptr, _lib.stream_ptr()
    rc = L.pir_trace(P(pk), n, P(o), P(d), P(lo), P(hi), P(wm), 5e-5, 16, 128, chunk, max_bisect,
                     P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(ws), ws.numel(), S)
    assert rc == 0
    torch.cuda.synchronize()
    before = torch.cuda.memory_allocated()
    rc = L.pir_trace_static(P(pk), n, P(o), P(d), P(lo), P(hi), P(wm), 5e-5, 16, 128, chunk, max_bisect,
                            P(b[0]), P(b[1]), P(b[2]), P(b[3]), P(status), S)
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.cuda.memory_allocated() == before
    return a, b, int(status.item())


def test_static_trace_matches_pir_trace_on_two_calls():
    net = sdf_net("sdf_full_bumpy_weights.npz")
    # first call: a 128 x 128 view; second call, smaller (40 x 40, other focal): stale workspace contents of the
    # first call must not leak into it
    for res, focal, chunk in ((128, 2.0, 5000), (40, 1.0, 0)):
        a, b, status = _trace_both(net, _rays(res, focal), chunk=chunk)
        assert status == 0
        assert int(a[0].sum()) > 0.05 * res * res          # the view has hits
        for x, y, name in zip(a, b, ("convergent", "points", "sdf", "distance")):
            assert torch.equal(x, y), (res, name, int((x != y).sum()))   # same kernels, same arithmetic: bit-exact


def test_static_trace_limits_and_empty():
    from pir_amd import _lib
    L = _lib.lib()
    net = sdf_net("sdf_full.npz")
    P, S = _lib.ptr, _lib.stream_ptr()
    pk = net.packed().buf
    n = 65537
    z3, z1 = torch.zeros(n, 3, device="cuda"), torch.zeros(n, device="cuda")
    m = torch.zeros(n, dtype=torch.uint8, device="cuda")
    conv, pts = torch.ones(n, dtype=torch.uint8, device="cuda"), torch.ones(n, 3, device="cuda")
    sdf, dis = torch.ones(n, device="cuda"), torch.ones(n, device="cuda")
    args = lambda k, chunk: (P(pk), k, P(z3), P(z3.clone()), P(z1), P(z1.clone()), P(m), 5e-5, 16, 128, chunk, 12,  # noqa: E731
                             P(conv), P(pts), P(sdf), P(dis), None, S)
    assert L.pir_trace_static(*args(n, 0)) == 4              # beyond PIR_TRACE_STATIC_MAX_RAYS
    assert L.pir_trace_static(*args(65536, 1000)) == 4       # more than 64 chunks
    assert L.pir_trace_static(*args(0, 0)) == 0              # empty
    assert L.pir_trace_static(*args(65536, 0)) == 0          # the maximum itself (no work: all masked out)
    torch.cuda.synchronize()
    assert int(conv[:65536].sum()) == 0 and int(conv[65536]) == 1
    with pytest.raises(_lib.PirHipError, match="maximum"):
        _lib.check(4, "pir_trace_static")
