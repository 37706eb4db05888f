"""Host side of the fused shading-input kernels (pir_shade_inputs_* in include/pir_hip.h): reparam_points,
normal normalisation and light direction / distance of render_normal_and_color + render_fn
(src/renderer/raytracer.py:812-850, render_surface.py:121-135) as ONE differentiable node."""
import torch

from . import _lib


class UnitNormals(torch.Tensor):
    """Marker type for normals that are already n / (|n| + 1e-10): pir_amd's render_fn does not normalise them again.
    (A render_fn that does -- the reference's own -- only divides a unit vector by 1 + 1e-10.)  The marker does not
    survive any torch operation: whatever a caller computes from these normals is a plain Tensor again and is
    re-normalised by render_fn as the reference does."""

    @staticmethod
    def mark(t):
        return t.as_subclass(UnitNormals)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        out = super().__torch_function__(func, types, args, kwargs or {})
        strip = lambda o: o.as_subclass(torch.Tensor) if isinstance(o, UnitNormals) else o  # noqa: E731
        if isinstance(out, (tuple, list)):
            return type(out)(strip(o) for o in out)
        return strip(out)


def _f(t):
    return t.detach().contiguous().float()


class _ShadeInputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, sdf, normal, ray_d, origin, per_point, want_lightdir, reparam):
        P = points.shape[0]
        dev = points.device
        pts, nrm, org = _f(points), _f(normal), _f(origin.reshape(-1, 3) if per_point else origin.reshape(3))
        nhat = torch.empty(P, 3, dtype=torch.float32, device=dev)
        lightdir = torch.empty(P, 3, dtype=torch.float32, device=dev) if want_lightdir else None
        dist = torch.empty(P, 1, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().pir_shade_inputs_forward(P, _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(org), int(per_point),
                                                           _lib.ptr(nhat), _lib.ptr(lightdir), _lib.ptr(dist),
                                                           _lib.stream_ptr()), "pir_shade_inputs_forward")
        ctx.save_for_backward(pts, nrm, _f(ray_d), org)
        ctx.meta = (per_point, want_lightdir, reparam, origin.shape, sdf.shape if sdf is not None else None)
        p_out = pts.clone()       # the re-parameterised point: same value, its own node (gradient -> sdf)
        if lightdir is None:
            lightdir = pts.new_zeros(0)
            ctx.mark_non_differentiable(lightdir)
        return p_out, nhat, lightdir, dist

    @staticmethod
    @torch.autograd.function.once_differentiable     # raw kernels below: a second differentiation must raise, not be cut
    def backward(ctx, g_p, g_nhat, g_ld, g_dist):
        pts, nrm, ray_d, org = ctx.saved_tensors
        per_point, want_lightdir, reparam, origin_shape, sdf_shape = ctx.meta
        P = pts.shape[0]
        dev = pts.device
        opt = lambda g, shape: None if g is None else _f(g).reshape(shape)  # noqa: E731
        g_p, g_nhat, g_dist = opt(g_p, (P, 3)), opt(g_nhat, (P, 3)), opt(g_dist, (P,))
        g_ld = opt(g_ld, (P, 3)) if want_lightdir else None
        g_normal = torch.empty(P, 3, dtype=torch.float32, device=dev)
        need_sdf = reparam and ctx.needs_input_grad[1]
        need_org = ctx.needs_input_grad[4] and not per_point
        g_sdf = torch.empty(P, dtype=torch.float32, device=dev) if need_sdf else None
        g_org = torch.empty(P, 3, dtype=torch.float32, device=dev) if need_org else None
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().pir_shade_inputs_backward(
                P, _lib.ptr(pts), _lib.ptr(nrm), _lib.ptr(ray_d), _lib.ptr(org), int(per_point), _lib.ptr(g_nhat),
                _lib.ptr(g_ld), _lib.ptr(g_dist), _lib.ptr(g_p), _lib.ptr(g_normal), _lib.ptr(g_sdf), _lib.ptr(g_org),
                _lib.stream_ptr()), "pir_shade_inputs_backward")
        return (None, g_sdf.reshape(sdf_shape) if need_sdf else None, g_normal, None,
                g_org.sum(dim=0).reshape(origin_shape) if need_org else None, None, None, None)


def shade_inputs(points, sdf, normal, ray_d, ray_o, light_pos, is_training):
    """-> (points', n_hat, lightdir or None, distance [P,1]).  points' carries reparam_points' gradient to `sdf` when
    is_training; lightdir is None for the collocated light (light_pos None: the caller uses ray_d, distance is
    |points - ray_o|)."""
    per_point = light_pos is None
    origin = ray_o if per_point else light_pos
    p, nhat, lightdir, dist = _ShadeInputs.apply(points, sdf if is_training else None, normal, ray_d, origin, per_point,
                                                 not per_point, bool(is_training))
    return p, UnitNormals.mark(nhat), (None if per_point else lightdir), dist
