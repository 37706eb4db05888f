"""GGX point-light shader, drop-in for the reference's src/renderer/renderer_ggx.py.

`GGXColocatedRenderer(use_cuda).forward(light, distance, normal, lightdir, viewdir, diffuse_albedo,
specular_albedo, alpha, EDirectRadiance=True) -> {"diffuse_rgb","specular_rgb","rgb"}` (renderer_ggx.py:19,39,126).
On a CUDA/ROCm device forward and backward are the fused HIP kernels pir_ggx_forward / pir_ggx_backward; on CPU
the same formula is evaluated with PyTorch ops (host-logic tests only).
"""
import os

import numpy as np
import torch
import torch.nn as nn

from .. import _lib

_HERE = os.path.dirname(os.path.abspath(__file__))
_ETA = 1.48958738
_F0 = 0.03867


def _aligned(t):
    t = t.contiguous().float()
    if t.data_ptr() % 16:
        t = t.clone()
    return t


class _GgxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, light, distance, normal, lightdir, viewdir, kd, ks, alpha, t_rough, t_diff, edr):
        P = normal.shape[0]
        dev = normal.device
        light_rgb = light.numel() > 1
        lt = _aligned(light.reshape(P, 3) if light_rgb else light.reshape(1))
        args = [_aligned(distance.reshape(P)), _aligned(normal), _aligned(lightdir), _aligned(viewdir), _aligned(kd),
                _aligned(ks), _aligned(alpha.reshape(P))]
        outs = [torch.empty(P, 3, dtype=torch.float32, device=dev) for _ in range(3)]
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().pir_ggx_forward(
                P, _lib.ptr(lt), int(light_rgb), int(edr), *[_lib.ptr(a) for a in args], _lib.ptr(t_rough),
                t_rough.numel(), _lib.ptr(t_diff), t_diff.numel(), *[_lib.ptr(o) for o in outs], _lib.stream_ptr()),
                "pir_ggx_forward")
        ctx.save_for_backward(lt, *args, t_rough, t_diff)
        ctx.meta = (P, light_rgb, bool(edr), light.shape, distance.shape, alpha.shape)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_d, g_s, g_rgb):
        lt, dist, normal, ldir, vdir, kd, ks, alpha, t_rough, t_diff = ctx.saved_tensors
        P, light_rgb, edr, light_shape, dist_shape, alpha_shape = ctx.meta
        dev = normal.device
        zero = None

        def tot(a, b):
            nonlocal zero
            if a is None and b is None:
                if zero is None:
                    zero = torch.zeros(P, 3, dtype=torch.float32, device=dev)
                return zero
            if a is None:
                return _aligned(b)
            if b is None:
                return _aligned(a)
            return _aligned(a + b)

        gd, gs = tot(g_d, g_rgb), tot(g_s, g_rgb)
        g_light = torch.empty((P, 3) if light_rgb else (P,), dtype=torch.float32, device=dev)
        g_dist = torch.empty(P, dtype=torch.float32, device=dev)
        g_alpha = torch.empty(P, dtype=torch.float32, device=dev)
        g3 = [torch.empty(P, 3, dtype=torch.float32, device=dev) for _ in range(5)]  # normal, ldir, vdir, kd, ks
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().pir_ggx_backward(
                P, _lib.ptr(lt), int(light_rgb), int(edr), _lib.ptr(dist), _lib.ptr(normal), _lib.ptr(ldir),
                _lib.ptr(vdir), _lib.ptr(kd), _lib.ptr(ks), _lib.ptr(alpha), _lib.ptr(t_rough), t_rough.numel(),
                _lib.ptr(t_diff), t_diff.numel(), _lib.ptr(gd), _lib.ptr(gs), _lib.ptr(g_light), _lib.ptr(g_dist),
                *[_lib.ptr(g) for g in g3], _lib.ptr(g_alpha), _lib.stream_ptr()), "pir_ggx_backward")
        gl = g_light.reshape(light_shape) if light_rgb else g_light.sum().reshape(light_shape)
        return (gl, g_dist.reshape(dist_shape), g3[0], g3[1], g3[2], g3[3], g3[4], g_alpha.reshape(alpha_shape),
                None, None, None)


def smithG1(cosTheta, alpha):
    """Smith shadowing term of one direction (renderer_ggx.py:8-12)."""
    root = alpha * (torch.sqrt(1.0 - cosTheta * cosTheta) / (cosTheta + 1e-10))
    return 2.0 / (1.0 + torch.hypot(root, torch.ones_like(root)))


def G_fn(dot_wi_n, dot_wo_n, alpha):
    """renderer_ggx.py:14-15."""
    return smithG1(dot_wi_n, alpha) * smithG1(dot_wo_n, alpha)


_smith_g1 = smithG1


class GGXColocatedRenderer(nn.Module):
    def __init__(self, use_cuda=False):
        super().__init__()
        self.MTS_TRANS = torch.from_numpy(np.load(os.path.join(_HERE, "ggx", "mts_rough_trans.npy")))   # 5000
        self.MTS_DIFF_TRANS = torch.from_numpy(np.load(os.path.join(_HERE, "ggx", "mts_diff_trans.npy")))  # 50
        self.num_theta_samples, self.num_alpha_samples = 100, 50
        if use_cuda:
            self.MTS_TRANS = self.MTS_TRANS.cuda()
            self.MTS_DIFF_TRANS = self.MTS_DIFF_TRANS.cuda()

    def _tables_on(self, device):
        if self.MTS_TRANS.device != device:
            self.MTS_TRANS = self.MTS_TRANS.to(device)
            self.MTS_DIFF_TRANS = self.MTS_DIFF_TRANS.to(device)
        return self.MTS_TRANS, self.MTS_DIFF_TRANS

    def forward(self, light, distance, normal, lightdir, viewdir, diffuse_albedo, specular_albedo, alpha,
                EDirectRadiance=True):
        if normal.is_cuda:
            t_rough, t_diff = self._tables_on(normal.device)
            shp = normal.shape
            P = normal.numel() // 3
            if not torch.is_tensor(light):
                light = torch.tensor(float(light), device=normal.device)
            bc = lambda t, w: t.expand(*shp[:-1], w).reshape(P, w)  # noqa: E731
            d, s, r = _GgxFn.apply(light.to(normal.device), bc(distance, 1), normal.reshape(P, 3),
                                   bc(lightdir, 3), bc(viewdir, 3), bc(diffuse_albedo, 3), bc(specular_albedo, 3),
                                   bc(alpha, 1), t_rough, t_diff, EDirectRadiance)
            return {"diffuse_rgb": d.reshape(shp), "specular_rgb": s.reshape(shp), "rgb": r.reshape(shp)}
        return self._forward_torch(light, distance, normal, lightdir, viewdir, diffuse_albedo, specular_albedo, alpha,
                                   EDirectRadiance)

    def _forward_torch(self, light, distance, normal, lightdir, viewdir, kd, ks, alpha, edr):
        inten = light / (distance * distance + 1e-10) if edr else light
        c_o = (viewdir * normal).sum(-1, keepdim=True).clamp(1e-5, 0.99999)
        c_i = (lightdir * normal).sum(-1, keepdim=True).clamp(1e-5, 0.99999)
        half = nn.functional.normalize(viewdir + lightdir, dim=-1, eps=1e-10)
        c_h = (half * normal).sum(-1, keepdim=True).clamp(1e-5, 0.99999)
        a = alpha.clamp(min=1e-4)
        ch2 = c_h * c_h
        root = ch2 + (1.0 - ch2) / (a * a + 1e-10)
        D = 1.0 / (np.pi * a * a * root * root + 1e-10)
        G = _smith_g1(c_i, a) * _smith_g1(c_o, a)
        spec = inten * ks * _F0 * D * G / (4.0 * c_i + 1e-10)
        tr, td = self._tables_on(normal.device)
        ty = torch.floor((a / 4.0) ** 0.25 * self.num_alpha_samples).long()
        ii = (ty * 100 + torch.floor(c_i ** 0.25 * 100).long()).clamp(0, tr.numel() - 1)
        io = (ty * 100 + torch.floor(c_o ** 0.25 * 100).long()).clamp(0, tr.numel() - 1)
        T12, T21 = tr[ii].clamp(0, 1), tr[io].clamp(0, 1)
        Fdr = (1.0 - td[ty.clamp(0, td.numel() - 1)]).clamp(0, 1)
        diff = inten * (kd / (1.0 - Fdr + 1e-10) / np.pi) * c_o * T12 * T21 * (1.0 / (_ETA * _ETA))
        return {"diffuse_rgb": diff, "specular_rgb": spec, "rgb": diff + spec}
