"""CPU oracle for the PIR per-ray surface-rendering hot path.

TEST INFRASTRUCTURE ONLY.  This module restates, in plain functional
torch-on-CPU code, the algorithm of the reference's hot path so that the HIP
kernels can be checked against it.  Only tests/, __graft_entry__.smoke() and
bench.py's `cpu_baseline` leg may import it; nothing under pir_amd/ does.

Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py)
against golden vectors captured from the reference's own modules by
tools/gen_golden.py (committed under tests/golden/).  The arithmetic that lives
in third-party code (GEMM, softplus, sigmoid, cumprod, autograd) is PyTorch's;
the reference pins no torch version, the fixtures were made with torch 2.10.0.

Each function cites the reference file:line (relative to the reference root)
whose semantics it follows.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

Tensor = torch.Tensor

# ----------------------------------------------------------------------------
# Positional encoding + weight-normed MLPs
# ----------------------------------------------------------------------------


def posenc(x: Tensor, n_freq: int) -> Tensor:
    """[x | sin(2^0 x) | cos(2^0 x) | ... | sin(2^(L-1) x) | cos(2^(L-1) x)].

    src/models/embedder.py:20-41 (include_input, log-sampled bands, sin then cos per band).
    """
    if n_freq <= 0:
        return x
    feats = [x]
    bands = 2.0 ** torch.linspace(0.0, n_freq - 1, n_freq)
    for k in range(n_freq):
        feats.append(torch.sin(x * bands[k]))
        feats.append(torch.cos(x * bands[k]))
    return torch.cat(feats, dim=-1)


def effective_weight(g: Tensor, v: Tensor) -> Tensor:
    """Old-style weight_norm, dim=0: W[o,:] = g[o] * v[o,:] / ||v[o,:]||  (src/network.py:75).

    torch._weight_norm is the op the reference's weight_norm hook itself evaluates; using it keeps the
    oracle's weights bit-identical to the reference's (the tracer is sensitive to 1-ulp weight changes).
    """
    if hasattr(torch, "_weight_norm"):
        return torch._weight_norm(v, g, 0)
    return v * (g / v.norm(dim=1, keepdim=True))


def mlp_layers(state: Dict[str, Tensor], prefix: str = "") -> List[Tuple[Tensor, Tensor]]:
    """Collect (W_eff, b) for lin0..linN from a reference-format state dict."""
    layers = []
    l = 0
    while f"{prefix}lin{l}.bias" in state:
        b = state[f"{prefix}lin{l}.bias"]
        if f"{prefix}lin{l}.weight_g" in state:
            w = effective_weight(state[f"{prefix}lin{l}.weight_g"], state[f"{prefix}lin{l}.weight_v"])
        else:
            w = state[f"{prefix}lin{l}.weight"]
        layers.append((w, b))
        l += 1
    return layers


def softplus100(z: Tensor) -> Tensor:
    """nn.Softplus(beta=100): log1p(exp(100 z))/100, identity where 100 z > 20 (src/network.py:79)."""
    return torch.nn.functional.softplus(z, beta=100.0, threshold=20.0)


def sdf_forward(layers: Sequence[Tuple[Tensor, Tensor]], x: Tensor, *, multires: int = 6,
                skip_in: Sequence[int] = (4,), scale: float = 1.0) -> Tensor:
    """SDFNetwork.forward: [M,3] -> [M,d_out] = [sdf/scale | feature].  src/network.py:81-97."""
    e = posenc(x * scale, multires)
    h = e
    n = len(layers)
    for l, (w, b) in enumerate(layers):
        if l in skip_in:
            h = torch.cat([h, e], dim=-1) / np.sqrt(2)
        h = h @ w.t() + b
        if l < n - 1:
            h = softplus100(h)
    return torch.cat([h[..., :1] / scale, h[..., 1:]], dim=-1)


def sdf_get_all(layers, x: Tensor, *, is_training: bool, **kw) -> Tuple[Tensor, Tensor, Tensor]:
    """value, feature, d(value)/dx via autograd; graph kept iff is_training.  src/network.py:125-142."""
    with torch.enable_grad():
        if not x.requires_grad:
            x.requires_grad_(True)
        out = sdf_forward(layers, x, **kw)
        y, feat = out[..., :1], out[..., 1:]
        (grad,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=is_training, retain_graph=is_training)
    if not is_training:
        return y.detach(), feat.detach(), grad.detach()
    return y, feat, grad


def sdf_gradient(layers, x: Tensor, **kw) -> Tensor:
    """SDFNetwork.gradient (always create_graph).  src/network.py:108-123."""
    return sdf_get_all(layers, x, is_training=True, **kw)[2]


# ----------------------------------------------------------------------------
# Camera / rays
# ----------------------------------------------------------------------------


class CameraRays:
    """Ray generation of Camera (src/renderer/raytracer.py:277-338)."""

    def __init__(self, W: int, H: int, K: Tensor, W2C: Tensor):
        self.W, self.H = int(W), int(H)
        self.K, self.W2C = K, W2C
        self.K_inv = torch.inverse(K)
        self.C2W = torch.inverse(W2C)

    def pixel_centers(self) -> Tensor:
        """:335-338 -- (u+0.5, v+0.5), u fastest."""
        v, u = torch.meshgrid(torch.arange(self.H, dtype=torch.float32), torch.arange(self.W, dtype=torch.float32),
                              indexing="ij")
        return torch.stack([u, v], dim=-1) + 0.5

    def rays(self, uv: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
        """:305-326 -- returns ray_o, unit ray_d, pre-normalisation length."""
        shp = list(uv.shape[:-1])
        p = torch.cat([uv.reshape(-1, 2), torch.ones(uv.numel() // 2, 1)], dim=-1)
        d = (p @ self.K_inv[:3, :3].t()) @ self.C2W[:3, :3].t()
        d = d.reshape(shp + [3])
        nrm = d.norm(dim=-1)
        d = d / nrm.unsqueeze(-1)
        o = self.C2W[:3, 3].expand(shp + [3])
        return o, d, nrm

    def light_pos(self, offset: Tensor) -> Tensor:
        """:297-303 -- camera-frame offset -> world position."""
        return offset @ self.C2W[:3, :3].t() + self.C2W[:3, 3].unsqueeze(0)

    def project(self, points: Tensor) -> Tensor:
        """:340-356 -- world points -> (u, v)."""
        p = torch.cat([points.reshape(-1, 3), torch.ones(points.numel() // 3, 1)], dim=1)
        q = (p @ self.W2C.t()) @ self.K.t()
        return (q[:, :2] / q[:, 2:3]).reshape(list(points.shape[:-1]) + [2])


def intersect_sphere(ray_o: Tensor, ray_d: Tensor, r: float = 1.0):
    """src/renderer/raytracer.py:253-273."""
    d1 = -(ray_d * ray_o).sum(-1) / (ray_d * ray_d).sum(-1)
    p = ray_o + d1.unsqueeze(-1) * ray_d
    tmp = r * r - (p * p).sum(-1)
    hit = tmp > 0.0
    d2 = torch.sqrt(tmp.clamp(min=0.0)) / ray_d.norm(dim=-1)
    return hit, (d1 - d2).clamp(min=0.0), d1 + d2


# ----------------------------------------------------------------------------
# Tracer
# ----------------------------------------------------------------------------


class Tracer:
    """RayTracer restated (src/renderer/raytracer.py:28-250)."""

    def __init__(self, sdf_threshold=5.0e-5, sphere_tracing_iters=16, n_steps=128, max_num_pts=200000):
        self.thr, self.iters, self.n_steps, self.max_num_pts = sdf_threshold, sphere_tracing_iters, n_steps, max_num_pts

    # :126-170
    def sphere_trace(self, sdf: Callable, ray_o, ray_d, min_dis, max_dis, work_mask):
        unfinished = work_mask.clone()
        acc = min_dis.clone()
        pts = ray_o + ray_d * acc.unsqueeze(-1)
        val = sdf(pts)
        it = 0
        counts = []
        while True:
            unfinished = unfinished & (val.abs() > self.thr) & (acc < max_dis)
            counts.append(int(unfinished.sum()))
            if it == self.iters or counts[-1] == 0:
                break
            it += 1
            step = val[unfinished]
            acc[unfinished] = acc[unfinished] + step
            pts[unfinished] = pts[unfinished] + ray_d[unfinished] * step.unsqueeze(-1)
            val[unfinished] = sdf(pts[unfinished])
        conv = work_mask & ~unfinished & (val.abs() <= self.thr) & (acc < max_dis)
        self.last_counts = counts
        return conv, unfinished, pts, val, acc

    # :229-250
    def bisect(self, sdf, f_lo, f_hi, d_lo, d_hi, ray_o, ray_d):
        work = (f_lo > 0) & (f_hi < 0)
        d_mid = (d_lo + d_hi) / 2.0
        self.last_bisect_iters = 0
        while work.any():
            f_mid = sdf(ray_o + ray_d * d_mid.unsqueeze(-1))
            pos = f_mid > 0
            neg = ~pos  # f_mid <= 0
            d_lo = torch.where(pos, d_mid, d_lo)
            f_lo = torch.where(pos, f_mid, f_lo)
            d_hi = torch.where(neg, d_mid, d_hi)
            f_hi = torch.where(neg, f_mid, f_hi)
            d_mid = (d_lo + d_hi) / 2.0
            work = work & ((d_hi - d_lo) > 2 * self.thr)
            self.last_bisect_iters += 1
        p = ray_o + ray_d * d_mid.unsqueeze(-1)
        return p, d_mid, sdf(p)

    # :172-227
    def sample_and_bisect(self, sdf, ray_o, ray_d, min_dis, max_dis):
        n = self.n_steps
        t = torch.linspace(0, 1, steps=n).float().view(1, n)
        z = min_dis.unsqueeze(-1) + t * (max_dis.unsqueeze(-1) - min_dis.unsqueeze(-1))
        pts = ray_o.unsqueeze(-2) + ray_d.unsqueeze(-2) * z.unsqueeze(-1)
        vals = torch.cat([sdf(c) for c in torch.split(pts.reshape(-1, 3), self.max_num_pts, dim=0)], 0).reshape(-1, n)
        out_p = torch.zeros_like(ray_d)
        out_s = torch.zeros_like(min_dis)
        out_d = torch.zeros_like(min_dis)
        score = torch.sign(vals) * torch.arange(n, 0, -1).float().reshape(1, n)
        best, idx = torch.min(score, dim=-1)  # first negative sample (largest weight among negatives)  (:192-197)
        self.last_min_idx, self.last_min_negative = idx, best < 0.0      # the path's index work, kept for the tests
        found = (best < 0.0) & (idx >= 1)
        if found.any():
            i = idx[found].unsqueeze(-1)
            zf, vf = z[found], vals[found]
            p, d, s = self.bisect(sdf, vf.gather(-1, i - 1).squeeze(-1), vf.gather(-1, i).squeeze(-1),
                                  zf.gather(-1, i - 1).squeeze(-1), zf.gather(-1, i).squeeze(-1),
                                  ray_o[found], ray_d[found])
            out_p[found], out_s[found], out_d[found] = p, s, d
        return found, out_p, out_s, out_d

    # :56-124
    @torch.no_grad()
    def trace(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask) -> Dict[str, Tensor]:
        conv, unf, pts, val, acc = self.sphere_trace(sdf, ray_o, ray_d, min_dis, max_dis, work_mask)
        if unf.sum() > 0:
            outside = (val[unf] > 0.0).float()
            lo = outside * acc[unf] + (1.0 - outside) * min_dis[unf]
            hi = outside * max_dis[unf] + (1.0 - outside) * acc[unf]
            f, p, s, d = self.sample_and_bisect(sdf, ray_o[unf], ray_d[unf], lo, hi)
            conv[unf], pts[unf], val[unf], acc[unf] = f, p, s, d
        return {"convergent_mask": conv, "points": pts, "sdf": val, "distance": acc}


@torch.no_grad()
def raytrace_camera(cam: CameraRays, layers, tracer: Tracer, max_num_rays: int = 200000, **kw) -> Dict[str, Tensor]:
    """raytrace_pixels + the trace part of raytrace_camera (fill_holes=False, detect_edges=False).

    src/renderer/raytracer.py:405-484, :618-628.
    """
    uv = cam.pixel_centers()
    ray_o, ray_d, ray_n = cam.rays(uv)
    sdf = lambda x: sdf_forward(layers, x, **kw)[..., 0]  # noqa: E731  (:426)
    parts: Dict[str, List[Tensor]] = {}
    ro, rd, rn = ray_o.reshape(-1, 3), ray_d.reshape(-1, 3), ray_n.reshape(-1)
    for s in range(0, ro.shape[0], max_num_rays):
        o, d, nn_ = ro[s:s + max_num_rays], rd[s:s + max_num_rays], rn[s:s + max_num_rays]
        hit, lo, hi = intersect_sphere(o, d, 1.0)
        res = tracer.trace(sdf, o, d, lo, hi, hit)
        res["depth"] = res["distance"] / nn_
        for k, v in res.items():
            parts.setdefault(k, []).append(v)
    out = {}
    for k, v in parts.items():
        t = torch.cat(v, 0)
        out[k] = t.reshape(cam.H, cam.W, 3) if t.dim() == 2 else t.reshape(cam.H, cam.W)
    out["depth"] = out["depth"] * out["convergent_mask"].float()
    out.update(uv=uv, ray_o=ray_o, ray_d=ray_d, ray_d_norm=ray_n)
    return out


def reparam_points(points: Tensor, grads: Tensor, dirs: Tensor, sdf_vals: Tensor) -> Tensor:
    """src/renderer/raytracer.py:18-25."""
    dot = (grads * dirs).sum(-1, keepdim=True).clamp(min=1e-4)
    return points - dirs / dot * (sdf_vals - sdf_vals.detach())


# ----------------------------------------------------------------------------
# Light visibility march
# ----------------------------------------------------------------------------


def light_visibility(light_pos: Tensor, points: Tensor, layers, inv_s: Tensor, n_samples: int = 128,
                     is_training: bool = True, **kw) -> Tensor:
    """src/renderer/raytracer.py:669-729."""
    bs = points.shape[0]
    delta = points - light_pos
    length = delta.norm(dim=-1, keepdim=True)
    d = delta / length
    discard = 10
    t = torch.linspace(0, 1, n_samples)[:-discard]
    n = n_samples - discard
    z = t.view(1, n) * length
    dz = torch.cat([z[:, 1:] - z[:, :-1], torch.full((bs, 1), 1.0 / n)], dim=-1)
    mid = z + 0.5 * dz
    x = (light_pos + d[:, None, :] * mid[:, :, None]).reshape(-1, 3)
    dirs = d[:, None, :].expand(bs, n, 3).reshape(-1, 3)
    sdf, _, g = sdf_get_all(layers, x, is_training=is_training, **kw)
    cos = (dirs * g).sum(-1, keepdim=True)
    cos = -torch.relu(-cos)
    half = cos * dz.reshape(-1, 1) * 0.5
    k = inv_s.reshape(1, 1)
    prev_cdf = torch.sigmoid((sdf - half) * k)
    next_cdf = torch.sigmoid((sdf + half) * k)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).reshape(bs, n).clip(0.0, 1.0)
    trans = torch.cumprod(torch.cat([torch.ones(bs, 1), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    vis = (1.0 - (alpha * trans).sum(-1, keepdim=True)).reshape(bs)
    return vis if is_training else vis.clone().detach()


# ----------------------------------------------------------------------------
# GGX point-light shader
# ----------------------------------------------------------------------------

_ETA = 1.48958738
_F0 = 0.03867


def _smith_g1(c: Tensor, a: Tensor) -> Tensor:
    """src/renderer/renderer_ggx.py:8-12."""
    root = a * (torch.sqrt(1.0 - c * c) / (c + 1e-10))
    return 2.0 / (1.0 + torch.hypot(root, torch.ones_like(root)))


def ggx_table_bins(c_i: Tensor, c_o: Tensor, a: Tensor, n_rough: int, n_diff: int):
    """The shader's integer work (src/renderer/renderer_ggx.py:94-99, :103-104, :115-116): bins of the rough-transmittance
    table for the incident and the outgoing cosine, and of the diffuse table; cosines and roughness already clamped."""
    ty = torch.floor((a / 4.0) ** 0.25 * 50).long()
    ix_i = (ty * 100 + torch.floor(c_i ** 0.25 * 100).long()).clamp(0, n_rough - 1)
    ix_o = (ty * 100 + torch.floor(c_o ** 0.25 * 100).long()).clamp(0, n_rough - 1)
    return ix_i, ix_o, ty.clamp(0, n_diff - 1)


def ggx_shade(light, distance, normal, lightdir, viewdir, kd, ks, alpha, tables: Tuple[Tensor, Tensor],
              direct_radiance: bool = True) -> Dict[str, Tensor]:
    """GGXColocatedRenderer.forward (src/renderer/renderer_ggx.py:39-127)."""
    t_rough, t_diff = tables
    inten = light / (distance * distance + 1e-10) if direct_radiance else light
    lo, hi = 0.00001, 0.99999
    c_o = (viewdir * normal).sum(-1, keepdim=True).clamp(lo, hi)
    c_i = (lightdir * normal).sum(-1, keepdim=True).clamp(lo, hi)
    h = torch.nn.functional.normalize(viewdir + lightdir, dim=-1, eps=1e-10)
    c_h = (h * normal).sum(-1, keepdim=True).clamp(lo, hi)
    a = alpha.clamp(min=0.0001)
    ch2 = c_h * c_h
    root = ch2 + (1.0 - ch2) / (a * a + 1e-10)
    D = 1.0 / (np.pi * a * a * root * root + 1e-10)
    G = _smith_g1(c_i, a) * _smith_g1(c_o, a)
    spec = inten * ks * _F0 * D * G / (4.0 * c_i + 1e-10)
    ix_i, ix_o, ix_d = ggx_table_bins(c_i, c_o, a, t_rough.numel(), t_diff.numel())
    T12 = t_rough[ix_i].clamp(0.0, 1.0)
    T21 = t_rough[ix_o].clamp(0.0, 1.0)
    Fdr = (1.0 - t_diff[ix_d]).clamp(0.0, 1.0)
    diff = inten * (kd / (1.0 - Fdr + 1e-10) / np.pi) * c_o * T12 * T21 * (1.0 / (_ETA * _ETA))
    return {"diffuse_rgb": diff, "specular_rgb": spec, "rgb": diff + spec}


def load_ggx_tables(directory: str) -> Tuple[Tensor, Tensor]:
    return (torch.from_numpy(np.load(f"{directory}/mts_rough_trans.npy")),
            torch.from_numpy(np.load(f"{directory}/mts_diff_trans.npy")))


# ----------------------------------------------------------------------------
# Material MLPs + render_fn
# ----------------------------------------------------------------------------


def rendering_mlp(layers, cfg: dict, points, normals, view_dirs, geo_f, semantic_f=None) -> Tensor:
    """RenderingNetwork.forward (src/network.py:217-249)."""
    mr, mrv = cfg.get("multires", 0), cfg.get("multires_view", 0)
    p = posenc(points, mr) if mr > 0 else points
    if cfg["mode"] == "idr":
        v = posenc(view_dirs, mrv) if mrv > 0 else view_dirs
        inp = torch.cat([p, v, normals, geo_f], dim=-1)
    else:  # no_view_dir
        inp = torch.cat([p, normals, geo_f, semantic_f], dim=-1)
    h = inp
    n = len(layers)
    skip = tuple(cfg.get("skip_in", ()))
    for l, (w, b) in enumerate(layers):
        if l in skip:
            h = torch.cat([h, inp], dim=-1) / np.sqrt(2)
        h = h @ w.t() + b
        if l < n - 1:
            h = torch.relu(h)
    h = cfg.get("output_scale", 1.0) * (h + cfg.get("output_bias", 0.0))
    if cfg.get("squeeze_out", True):
        h = cfg.get("squeeze_out_scale", 1.0) * torch.sigmoid(h)
    return h


def dino_mlp(layers, cfg: dict, points) -> Tensor:
    """DINONetwork.forward with points only (src/network.py:436-467)."""
    h = posenc(points, cfg.get("multires", 0))
    n = len(layers)
    for l, (w, b) in enumerate(layers):
        h = h @ w.t() + b
        if l < n - 1:
            h = torch.relu(h)
    return cfg.get("output_scale", 1.0) * (h + cfg.get("output_bias", 0.0))


def materials(mats: dict, points, normals, geo_f, dino_f, roughness_warmup=False, is_metal=False):
    """get_materials without the smoothness branch (render_surface.py:51-86)."""
    kd = rendering_mlp(mats["diffuse_albedo_network"]["layers"], mats["diffuse_albedo_network"]["cfg"],
                       points, normals, -normals, geo_f).abs()[..., [2, 1, 0]]
    ks = rendering_mlp(mats["specular_albedo_network"]["layers"], mats["specular_albedo_network"]["cfg"],
                       points, normals, None, geo_f, dino_f).abs()
    if not is_metal:
        ks = ks.mean(dim=-1, keepdim=True).expand_as(ks)
    if roughness_warmup:
        rough = torch.ones(ks.shape[0], 1) * 0.10
    else:
        rough = rendering_mlp(mats["specular_roughness_network"]["layers"], mats["specular_roughness_network"]["cfg"],
                              points, normals, None, geo_f, dino_f).abs() + 0.01
    return kd, ks, rough


def render_fn(interior_mask, mats, tables, ray_o, ray_d, points, normals, geo_f, dino_f, light, lightdir=None,
              distance=None, roughness_warmup=False, direct_radiance=True) -> Dict[str, Tensor]:
    """render_surface.py:89-176 (smoothness grads are zeros: args.smoothness_weight == 0 path)."""
    n = interior_mask.shape[0]

    def pad(v, width):
        buf = torch.zeros(n, width, dtype=v.dtype) if width > 0 else torch.zeros(n, dtype=v.dtype)
        if interior_mask.any():
            buf[interior_mask] = v
        return buf

    if interior_mask.any():
        nrm = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
        kd, ks, rough = materials(mats, points, nrm, geo_f, dino_f, roughness_warmup)
        if lightdir is None:
            lightdir = ray_d
        if distance is None:
            distance = (points - ray_o).norm(dim=-1, keepdim=True)
        sh = ggx_shade(light, distance, nrm, -lightdir, -ray_d, kd, ks, rough, tables, direct_radiance)
        res = {"color": pad(sh["rgb"], 3), "diffuse_color": pad(sh["diffuse_rgb"], 3),
               "specular_color": pad(sh["specular_rgb"], 3), "diffuse_albedo": pad(kd, 3),
               "specular_albedo": pad(ks, 3), "specular_roughness": pad(rough.squeeze(-1), 0), "normal": pad(nrm, 3)}
    else:
        z3 = torch.zeros(n, 3)
        res = {"color": z3, "diffuse_color": z3.clone(), "specular_color": z3.clone(), "diffuse_albedo": z3.clone(),
               "specular_albedo": z3.clone(), "specular_roughness": torch.zeros(n), "normal": z3.clone()}
    res["specular_grads"] = torch.zeros(n, 3)
    res["roughness_grads"] = torch.zeros(n, 3)
    return res


# ----------------------------------------------------------------------------
# Whole-camera step (the bench's CPU baseline and the end-to-end parity case)
# ----------------------------------------------------------------------------


def render_view(cam: CameraRays, sdf_layers, mats, dino, tables, light, *, light_offset: Optional[Tensor],
                inv_s: Tensor, n_vis_samples: int = 128, is_training: bool = True,
                vis_chunk: int = 512) -> Dict[str, Tensor]:
    """render_camera with fill_holes=False, handle_edges=False, indirect_warmup=True.

    src/renderer/raytracer.py:1198-1275 -> raytrace_camera (:618), render_light_visibility (:732),
    render_normal_and_color (:780).
    """
    res = raytrace_camera(cam, sdf_layers, Tracer())
    mask = res["convergent_mask"].reshape(-1)
    pts = res["points"].reshape(-1, 3)[mask]
    ray_d = res["ray_d"].reshape(-1, 3)[mask]
    ray_o = res["ray_o"].reshape(-1, 3)[mask]
    light_pos = None
    if light_offset is not None:
        light_pos = cam.light_pos(light_offset)
        vis_parts = []
        full_pts = res["points"].reshape(-1, 3)
        for s in range(0, full_pts.shape[0], vis_chunk):  # :746-771 chunks the *image*, not the hit list
            m = mask[s:s + vis_chunk]
            buf = torch.zeros(m.shape[0])
            if m.any():
                buf[m] = light_visibility(light_pos, full_pts[s:s + vis_chunk][m], sdf_layers, inv_s,
                                          n_samples=n_vis_samples, is_training=is_training)
            vis_parts.append(buf)
        res["vis"] = torch.cat(vis_parts).reshape(cam.H, cam.W)
    if mask.any():
        sdf, feat, nrm = sdf_get_all(sdf_layers, pts, is_training=is_training)
        dino_f = dino_mlp(dino["layers"], dino["cfg"], pts).detach()
        p = reparam_points(pts, nrm.detach(), -ray_d.detach(), sdf) if is_training else pts
        lightdir = distance = None
        with torch.set_grad_enabled(is_training):
            if light_pos is not None:
                lightdir = torch.nn.functional.normalize(p - light_pos, eps=1e-10)
                distance = (p - light_pos).norm(dim=-1, keepdim=True)
            out = render_fn(mask, mats, tables, ray_o, ray_d, p, nrm, feat, dino_f, light, lightdir, distance)
    else:
        out = render_fn(mask, mats, tables, None, None, None, None, None, None, light)
    for k, v in out.items():
        res[k] = v.reshape(cam.H, cam.W, 3) if v.dim() == 2 else v.reshape(cam.H, cam.W)
    return res


def step_loss(res: Dict[str, Tensor]) -> Tensor:
    """Loss used by bench/tests: sum(color * vis) + eikonal on the returned normals."""
    color = res["color"]
    if "vis" in res:
        color = color * res["vis"][..., None]
    m = res["convergent_mask"]
    return color.sum() + ((res["normal"][m].norm(dim=-1) - 1.0) ** 2).sum()


# ----------------------------------------------------------------------------
# Next rows: IRON edge sampling and one-bounce indirect lighting
# ----------------------------------------------------------------------------


def sobel_magnitude(depth: Tensor) -> Tensor:
    """kornia.filters.sobel(depth[None, None]) as raytrace_camera calls it (src/renderer/raytracer.py:645).

    kornia is a third-party dependency that is absent here and unpinned in the reference (README.md:46);
    this restates its published algorithm: 3x3 Sobel kernels normalised by 8, replicate padding,
    sqrt(gx^2 + gy^2 + 1e-6).  PARITY UNPINNED for this function (no kornia output to compare with).
    """
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0
    x = torch.nn.functional.pad(depth[None, None], (1, 1, 1, 1), mode="replicate")
    g = torch.nn.functional.conv2d(x, torch.stack([kx, kx.t()])[:, None])[0]
    return torch.sqrt(g[0] * g[0] + g[1] * g[1] + 1e-6)


def raytrace_uv(cam: CameraRays, layers, tracer: Tracer, uv: Tensor, **kw) -> Dict[str, Tensor]:
    """raytrace_pixels for an arbitrary list of pixel positions (src/renderer/raytracer.py:405-484)."""
    ray_o, ray_d, ray_n = cam.rays(uv)
    sdf = lambda x: sdf_forward(layers, x, **kw)[..., 0]  # noqa: E731
    hit, lo, hi = intersect_sphere(ray_o, ray_d, 1.0)
    with torch.no_grad():
        res = tracer.trace(sdf, ray_o, ray_d, lo, hi, hit)
    res["depth"] = res["distance"] / ray_n
    res.update(uv=uv, ray_o=ray_o, ray_d=ray_d, ray_d_norm=ray_n)
    return res


def first_occurrence_unique(x: Tensor) -> Tuple[Tensor, Tensor]:
    """`unique` (src/renderer/raytracer.py:487-494): sorted unique values and, for each, the index of its first
    occurrence in x (the reference gets it from a scatter of the flipped index range, last write wins)."""
    vals, inv = torch.unique(x, return_inverse=True)
    first = torch.full((vals.shape[0],), x.shape[0], dtype=torch.long)
    for i in range(x.shape[0] - 1, -1, -1):
        first[inv[i]] = i
    return vals, first


@torch.no_grad()
def locate_edge_points(cam: CameraRays, start_points: Tensor, layers, mask: Tensor, max_step: int = 16,
                       step_size: float = 1e-3, dot_threshold: float = 5e-2, **kw) -> Dict[str, Tensor]:
    """Surface walk towards the silhouette (src/renderer/raytracer.py:497-589)."""
    cam_o = cam.C2W[:3, 3]
    pts = start_points[mask].reshape(-1, 3).clone()
    found = torch.zeros(pts.shape[0], dtype=torch.bool)
    if pts.shape[0] > 0:
        for it in range(max_step + 1):
            todo = ~found
            view = cam_o - pts[todo]
            view = view / (view.norm(dim=-1, keepdim=True) + 1e-10)
            sdf, _, nrm = sdf_get_all(layers, pts[todo], is_training=False, **kw)
            nrm = nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-10)
            dot = (nrm * view).sum(-1)
            keep_walking = dot.abs() > dot_threshold                      # :531
            found[todo] = ~keep_walking
            if it >= max_step or not (~found).any():
                break
            walk = nrm - view / dot.unsqueeze(-1)                          # :538
            walk = walk / (walk.norm(dim=-1, keepdim=True) + 1e-10)
            walk = walk - sdf * nrm                                        # :541
            idx = todo.nonzero()[:, 0][keep_walking]
            pts[idx] += (step_size * walk)[keep_walking]
    edge_points = pts[found]
    edge_mask = torch.zeros(cam.H, cam.W, dtype=torch.bool)
    edge_uv = torch.zeros(edge_points.shape[0], 2)
    pix = torch.zeros(0, dtype=torch.long)
    if found.any():
        edge_uv = cam.project(edge_points)
        ij = torch.floor(edge_uv).long()
        pix = ij[:, 1] * cam.W + ij[:, 0]
        ok = (pix < cam.H * cam.W) & (pix >= 0)                            # :569
        pix, edge_points, edge_uv = pix[ok], edge_points[ok], edge_uv[ok]
        if ok.any():
            pix, first = first_occurrence_unique(pix)
            edge_points, edge_uv = edge_points[first], edge_uv[first]
            edge_mask.view(-1)[pix] = True
    return {"edge_mask": edge_mask, "edge_points": edge_points, "edge_uv": edge_uv, "edge_pixel_idx": pix}


def blend_mlp(layers, cfg: dict, distance: Tensor, roughness: Tensor, dot_n_wo: Tensor) -> Tensor:
    """BlendNetwork.forward (src/network.py:344-367), no skip connections, no extra features."""
    h = torch.cat([distance, roughness, posenc(dot_n_wo, cfg.get("multires_view", 8))], dim=-1)
    n = len(layers)
    for l, (w, b) in enumerate(layers):
        h = h @ w.t() + b
        if l < n - 1:
            h = torch.relu(h)
    h = cfg.get("output_scale", 1.0) * (h + cfg.get("output_bias", 0.0))
    if cfg.get("squeeze_out", True):
        return cfg.get("squeeze_out_scale", 1.0) * torch.sigmoid(h)
    return torch.relu(h) * 2.0 * np.pi


def shade_traced(res: Dict[str, Tensor], sdf_layers, mats, dino, tables, light, *, light_pos: Optional[Tensor],
                 is_training: bool) -> Dict[str, Tensor]:
    """render_normal_and_color on a flat list of traced rays (src/renderer/raytracer.py:780-865)."""
    mask = res["convergent_mask"].reshape(-1)
    if not mask.any():
        return render_fn(mask, mats, tables, None, None, None, None, None, None, light)
    pts = res["points"].reshape(-1, 3)[mask]
    ray_d = res["ray_d"].reshape(-1, 3)[mask]
    ray_o = res["ray_o"].reshape(-1, 3)[mask]
    sdf, feat, nrm = sdf_get_all(sdf_layers, pts, is_training=is_training)
    dino_f = dino_mlp(dino["layers"], dino["cfg"], pts).detach()
    p = reparam_points(pts, nrm.detach(), -ray_d.detach(), sdf) if is_training else pts
    lightdir = distance = None
    with torch.set_grad_enabled(is_training):
        if light_pos is not None:
            lightdir = torch.nn.functional.normalize(p - light_pos, eps=1e-10)
            distance = (p - light_pos).norm(dim=-1, keepdim=True)
        out = render_fn(mask, mats, tables, ray_o, ray_d, p, nrm, feat, dino_f, light, lightdir, distance)
    out["dino_feature"] = torch.zeros(mask.shape[0], dino_f.shape[-1]).index_put((mask.nonzero()[:, 0],), dino_f)
    return out


def render_indirect(res: Dict[str, Tensor], sdf_layers, mats, dino, tables, tracer: Tracer, light, *,
                    is_training: bool) -> Tuple[Tensor, Tensor]:
    """render_indirect_lighting as render_camera calls it: light_pos=None, sec_is_training=False, diff_geo=True
    (src/renderer/raytracer.py:1003-1189).  Returns (indirect_color [H,W,3], indirect_blend_coef [H,W,1])."""
    H, W = res["convergent_mask"].shape
    pmask = res["convergent_mask"].reshape(-1)
    color = torch.zeros(H * W, 3)
    coef = torch.ones(H * W)
    if pmask.any():
        pts = res["points"].reshape(-1, 3)[pmask].clone()
        ray_o = res["ray_o"].reshape(-1, 3)[pmask]
        ray_d = res["ray_d"].reshape(-1, 3)[pmask]
        nrm = sdf_get_all(sdf_layers, pts, is_training=False)[2]
        nrm = nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-10)
        wo = (-ray_d * nrm).sum(-1, keepdim=True) * nrm * 2 + ray_d         # mirror direction, :1041
        wo = wo / (wo.norm(dim=-1, keepdim=True) + 1e-10)
        sdf = lambda x: sdf_forward(sdf_layers, x)[:, 0]  # noqa: E731
        with torch.no_grad():
            hit, lo, hi = intersect_sphere(pts, wo, 1.0)
            sec = tracer.trace(sdf, pts, wo, lo.clamp(0.001, 3.0), hi.clamp(0.001, 3.0), hit)
        m2 = sec["convergent_mask"]
        hit_pts = sec["points"][m2]
        # secondary shading, no gradients (:1075-1112): light at the camera, radiance falls off with distance^2
        ldir = hit_pts - ray_o[m2]
        hdist = ldir.norm(dim=-1, keepdim=True)
        ldir = ldir / (ldir.norm(dim=-1, keepdim=True) + 1e-10)
        h_sdf, h_feat, h_nrm = sdf_get_all(sdf_layers, hit_pts, is_training=False)
        h_dino = dino_mlp(dino["layers"], dino["cfg"], hit_pts).detach()
        with torch.no_grad():
            sec_out = render_fn(m2, mats, tables, pts[m2], wo[m2], hit_pts, h_nrm, h_feat, h_dino, light,
                                lightdir=ldir, distance=hdist, direct_radiance=True)
        radiance = sec_out["color"][m2]
        # primary shading lit by the secondary radiance from -wo (:1116-1144)
        p1, o1, d1, w1 = pts[m2], ray_o[m2], ray_d[m2], wo[m2]
        p_sdf, p_feat, p_nrm = sdf_get_all(sdf_layers, p1, is_training=is_training)
        p_dino = dino_mlp(dino["layers"], dino["cfg"], p1).detach()
        if is_training:
            p1 = reparam_points(p1, p_nrm.detach(), -d1.detach(), p_sdf)
        with torch.set_grad_enabled(is_training):
            out = render_fn(m2, mats, tables, o1, d1, p1, p_nrm, p_feat, p_dino, radiance, lightdir=-w1,
                            distance=None, direct_radiance=False)
            blend = torch.ones(m2.shape[0])
            if m2.any():
                rough = out["specular_roughness"][m2].reshape(-1, 1).detach()
                n1 = p_nrm.detach()
                n1 = n1 / (n1.norm(dim=-1, keepdim=True) + 1e-10)
                dot = (w1 * n1).sum(-1, keepdim=True)
                dist = (p1 - hit_pts).norm(dim=-1, keepdim=True)
                b = blend_mlp(mats["blend_network"]["layers"], mats["blend_network"]["cfg"], dist, rough, dot)
                blend = blend.index_put((m2.nonzero()[:, 0],), b.squeeze(-1))
        idx = pmask.nonzero()[:, 0]
        color = color.index_put((idx,), out["color"])
        coef = coef.index_put((idx,), blend)
    return color.reshape(H, W, 3), coef.reshape(H, W, 1)


def render_edge_pixels(res: Dict[str, Tensor], cam: CameraRays, sdf_layers, mats, dino, tables, tracer: Tracer,
                       light, *, light_pos: Optional[Tensor], is_training: bool) -> None:
    """Edge pixels = area-weighted mix of the colours either side of the silhouette
    (src/renderer/raytracer.py:872-951).  Updates color / normal / uv / points of `res` in place."""
    H, W = res["edge_mask"].shape
    e_pts, e_uv, e_idx = res["edge_points"], res["edge_uv"], res["edge_pixel_idx"]
    centre = torch.floor(e_uv) + 0.5
    e_sdf, _, e_grad = sdf_get_all(sdf_layers, e_pts, is_training=is_training)
    e_n = e_grad.detach() / (e_grad.detach().norm(dim=-1, keepdim=True) + 1e-10)
    if is_training:
        e_pts = reparam_points(e_pts, e_grad.detach(), e_n, e_sdf)
        e_uv = cam.project(e_pts)
    n2d = (e_n @ cam.W2C[:3, :3].t())[:, :2]
    n2d = n2d / (n2d.norm(dim=-1, keepdim=True) + 1e-10)
    radius = 0.707
    dot2d = ((e_uv - centre) * n2d).sum(-1)
    alpha = 2 * torch.arccos((dot2d / radius).clamp(0.0, 1.0))
    w_pos = (1.0 - (alpha - torch.sin(alpha)) / (2.0 * np.pi)).unsqueeze(-1)
    sides = []
    for sign in (-1.0, 1.0):                                                # positive side first (:905-906)
        tr = raytrace_uv(cam, sdf_layers, tracer, centre + sign * radius * n2d)
        sides.append((tr, shade_traced(tr, sdf_layers, mats, dino, tables, light, light_pos=light_pos,
                                       is_training=is_training)))
    (tr_p, pos), (tr_n, neg) = sides
    mix = lambda k: pos[k] * w_pos + neg[k] * (1.0 - w_pos)  # noqa: E731
    res["color"] = res["color"].reshape(-1, 3).index_put((e_idx,), mix("color")).reshape(H, W, 3)
    res["normal"] = res["normal"].reshape(-1, 3).index_put((e_idx,), e_grad).reshape(H, W, 3)
    res["edge_pos_neg_normal"] = torch.cat([pos["normal"][tr_p["convergent_mask"]],
                                            neg["normal"][tr_n["convergent_mask"]]], dim=0)
    res["uv"] = res["uv"].reshape(-1, 2).index_put((e_idx,), e_uv.detach()).reshape(H, W, 2)
    res["points"] = res["points"].reshape(-1, 3).index_put((e_idx,), e_pts.detach()).reshape(H, W, 3)
