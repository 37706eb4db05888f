"""torch.autograd.Function wrappers of the light-march kernels (pir_visibility_* in include/pir_hip.h).

light_visibility = per-point geometry (PyTorch, [bs] sized) -> `march_points` -> SDFNetwork.get_sdf_normal (HIP)
-> `composite`.  Everything sized [bs * n] is produced and consumed by HIP kernels only.
"""
import torch

from . import _lib


def _f(t):
    return t.detach().contiguous().float()


class _MarchPoints(torch.autograd.Function):
    @staticmethod
    def forward(ctx, light_pos, rays_d, length, n_samples):
        bs = rays_d.shape[0]
        n = n_samples - 10
        lp, d, L = _f(light_pos.reshape(3)), _f(rays_d), _f(length.reshape(bs))
        x = torch.empty(bs * n, 3, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.lib().pir_visibility_points(_lib.ptr(lp), _lib.ptr(d), _lib.ptr(L), bs, n_samples,
                                                        _lib.ptr(x), _lib.stream_ptr()), "pir_visibility_points")
        ctx.save_for_backward(d, L)
        ctx.meta = (n_samples, light_pos.shape, length.shape)
        return x

    @staticmethod
    def backward(ctx, g_x):
        d, L = ctx.saved_tensors
        n_samples, lp_shape, len_shape = ctx.meta
        bs = d.shape[0]
        g_lp = torch.empty(bs, 3, dtype=torch.float32, device=d.device)
        g_d = torch.empty(bs, 3, dtype=torch.float32, device=d.device)
        g_L = torch.empty(bs, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.lib().pir_visibility_points_backward(
                _lib.ptr(_f(g_x)), _lib.ptr(d), _lib.ptr(L), bs, n_samples, _lib.ptr(g_lp), _lib.ptr(g_d),
                _lib.ptr(g_L), _lib.stream_ptr()), "pir_visibility_points_backward")
        return g_lp.sum(dim=0).reshape(lp_shape), g_d, g_L.reshape(len_shape), None


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sdf, normal, rays_d, length, inv_s, n_samples):
        bs = rays_d.shape[0]
        s, nr, d, L, k = _f(sdf.reshape(-1)), _f(normal), _f(rays_d), _f(length.reshape(bs)), _f(inv_s.reshape(1))
        vis = torch.empty(bs, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.lib().pir_visibility_forward(_lib.ptr(s), _lib.ptr(nr), _lib.ptr(d), _lib.ptr(L),
                                                         _lib.ptr(k), bs, n_samples, _lib.ptr(vis),
                                                         _lib.stream_ptr()), "pir_visibility_forward")
        ctx.save_for_backward(s, nr, d, L, k)
        ctx.meta = (n_samples, sdf.shape, length.shape)
        return vis

    @staticmethod
    def backward(ctx, g_vis):
        s, nr, d, L, k = ctx.saved_tensors
        n_samples, sdf_shape, len_shape = ctx.meta
        bs = d.shape[0]
        g_s = torch.empty_like(s)
        g_n = torch.empty_like(nr)
        g_d = torch.empty(bs, 3, dtype=torch.float32, device=d.device)
        g_L = torch.empty(bs, dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.lib().pir_visibility_backward(
                _lib.ptr(_f(g_vis)), _lib.ptr(s), _lib.ptr(nr), _lib.ptr(d), _lib.ptr(L), _lib.ptr(k), bs, n_samples,
                _lib.ptr(g_s), _lib.ptr(g_n), _lib.ptr(g_d), _lib.ptr(g_L), _lib.stream_ptr()),
                "pir_visibility_backward")
        return g_s.reshape(sdf_shape), g_n, g_d, g_L.reshape(len_shape), None, None


def march_points(light_pos, rays_d, length, n_samples):
    """[bs*(n_samples-10), 3] sample positions between the light and each point."""
    return _MarchPoints.apply(light_pos, rays_d, length, n_samples)


def composite(sdf, normal, rays_d, length, inv_s, n_samples):
    """vis [bs] from the samples' sdf [bs*n] and normal [bs*n,3]."""
    return _Composite.apply(sdf, normal, rays_d, length, inv_s, n_samples)
