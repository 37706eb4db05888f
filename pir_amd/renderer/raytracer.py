"""Drop-in counterparts of the hot-path functions of the reference's src/renderer/raytracer.py.

Same names, arguments and result dictionaries as the reference (cited per function); on a CUDA/ROCm device the
work is done by the HIP kernels in libpir_hip.so:

  RayTracer.forward / raytrace_pixels  -> pir_camera_rays + pir_trace (sphere tracing, sampler, bisection)
  light_visibility                     -> pir_sdf_get_all (+ backward) for the 118/246 march samples per point
  render_normal_and_color              -> pir_sdf_get_all (+ backward), pir_ggx_forward/backward via render_fn

The per-image chunking arguments (max_num_rays, max_num_pts) are accepted for signature compatibility; the
kernels stream over any number of rays, so they only keep their one semantic effect (bisection rounds are
shared by the rays of one max_num_rays chunk, raytracer.py:234).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib

VERBOSE_MODE = False
dimension = 384
FUSED_EDGE_WALK = True      # A/B switch: False = one get_all launch + eager ops + a host sync per walk step
FUSED_SHADE_INPUTS = True   # A/B switch: False = the eager PyTorch expressions of the reference for a9 / shading inputs


def reparam_points(nondiff_points, nondiff_grads, nondiff_trgt_dirs, diff_sdf_vals):
    """Value-identical to the input points; gradient -dir/clamp(grad.dir, 1e-4) * d sdf (raytracer.py:18-25)."""
    dot = (nondiff_grads * nondiff_trgt_dirs).sum(dim=-1, keepdim=True).clamp(min=1e-4)
    return nondiff_points - nondiff_trgt_dirs / dot * (diff_sdf_vals - diff_sdf_vals.detach())


class SdfCallable:
    """`sdf(x) -> [...]` closure that still exposes the network, so RayTracer can run natively on it.
    (The reference builds `lambda x: sdf_network(x)[..., 0]`, raytracer.py:426, which hides the weights.)"""

    def __init__(self, sdf_network):
        self.sdf_network = sdf_network

    def __call__(self, x):
        return self.sdf_network.sdf(x)[..., 0]


_trace_ws = {}


def _workspace(device, nbytes):
    key = _lib.stream_key(device)      # one workspace per (device, stream): traces on different streams do not race
    buf = _trace_ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _trace_ws[key] = buf
    return buf


class RayTracer(nn.Module):
    """Sphere tracing + dense sampling + bisection (raytracer.py:28-250)."""

    def __init__(self, sdf_threshold=5.0e-5, sphere_tracing_iters=16, n_steps=128, max_num_pts=200000):
        super().__init__()
        self.sdf_threshold = sdf_threshold
        self.sphere_tracing_iters = sphere_tracing_iters
        self.n_steps = n_steps
        self.max_num_pts = max_num_pts
        # Bisection rounds the HIP path enqueues.  The reference halves until every bracket is below 2*threshold
        # (raytracer.py:229-250); a bracket starts at span/(n_steps-1), so ceil(log2(span/((n_steps-1) 2 thr))) rounds
        # finish it.  The count below covers spans up to 1024 (unit-sphere rays have span <= 2, the indirect bounce
        # <= 3); rounds with nothing left to do are skipped on the device, so the margin is free.  Should the bound
        # ever be hit the kernels raise a status word that is accumulated per device and checked by check_status().
        self.max_bisect = self.bisect_rounds_for(1024.0)
        self._status = {}      # (device index, stream) -> int32[1] tensor: OR of the status words of every trace there

    def bisect_rounds_for(self, span):
        """Rounds after which a bracket of span/(n_steps-1) is below 2*sdf_threshold, plus one; within the C ABI's 1..64."""
        import math
        first = float(span) / max(1, self.n_steps - 1)
        rounds = math.ceil(math.log2(max(first / (2.0 * self.sdf_threshold), 1.0))) + 1
        return int(min(64, max(1, rounds)))

    # ---------------------------------------------------------------- native path
    def _trace_hip(self, net, ray_o, ray_d, min_dis, max_dis, work_mask, chunk):
        shp = min_dis.shape
        dev = ray_o.device
        flat = lambda t, w: t.detach().reshape(-1, w).contiguous().float() if w else t.detach().reshape(-1).contiguous().float()  # noqa: E731
        o, d = flat(ray_o, 3), flat(ray_d, 3)
        lo, hi = flat(min_dis, 0), flat(max_dis, 0)
        wm = work_mask.detach().reshape(-1).contiguous().to(torch.uint8)
        n = lo.shape[0]
        conv = torch.empty(n, dtype=torch.uint8, device=dev)
        pts = torch.empty(n, 3, dtype=torch.float32, device=dev)
        sdf = torch.empty(n, dtype=torch.float32, device=dev)
        dis = torch.empty(n, dtype=torch.float32, device=dev)
        L = _lib.lib()
        chunk = int(chunk) if chunk else 0
        ws = _workspace(dev, L.pir_trace_workspace_bytes(n, chunk, self.max_bisect))
        with torch.cuda.device(dev):
            _lib.check(L.pir_trace(_lib.ptr(net.packed().buf), n, _lib.ptr(o), _lib.ptr(d), _lib.ptr(lo), _lib.ptr(hi),
                                   _lib.ptr(wm), float(self.sdf_threshold), int(self.sphere_tracing_iters),
                                   int(self.n_steps), chunk, int(self.max_bisect), _lib.ptr(conv), _lib.ptr(pts),
                                   _lib.ptr(sdf), _lib.ptr(dis), _lib.ptr(ws), ws.numel(), _lib.stream_ptr()),
                       "pir_trace")
        # accumulate the tracer status word (counter 33 of the workspace header) on the device: no host sync here
        # (one accumulator per stream, like the workspace: traces on two streams never update the same word)
        key = _lib.stream_key(dev)
        acc = self._status.get(key)
        if acc is None:
            acc = self._status[key] = torch.zeros(1, dtype=torch.int32, device=dev)
        acc.bitwise_or_(ws[:256].view(torch.int32)[33:34])
        return {"convergent_mask": conv.bool().reshape(shp), "points": pts.reshape(*shp, 3),
                "sdf": sdf.reshape(shp), "distance": dis.reshape(shp)}

    @torch.no_grad()
    def forward_static(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask, out, status=None, chunk=None):
        """`forward` for callers that must not allocate device memory (HIP graph capture, serving loops): runs
        pir_trace_static, whose workspace lives inside the library (<= 65 536 rays).  Inputs must already be
        contiguous fp32 / uint8 CUDA tensors; `out` holds the preallocated results: convergent_mask uint8 [n],
        points [n,3], sdf [n], distance [n]; `status`: optional int32 [1] device tensor (1 = bisection bound hit)."""
        net = sdf.sdf_network
        n = min_dis.numel()
        if n > 65536:
            raise _lib.PirHipError("forward_static: at most 65 536 rays per call (PIR_ERR_TOO_LARGE)")
        for t, dt in ((ray_o, torch.float32), (ray_d, torch.float32), (min_dis, torch.float32),
                      (max_dis, torch.float32), (work_mask, torch.uint8), (out["convergent_mask"], torch.uint8),
                      (out["points"], torch.float32), (out["sdf"], torch.float32), (out["distance"], torch.float32)):
            assert t.is_cuda and t.is_contiguous() and t.dtype == dt, "forward_static takes contiguous fp32/uint8 CUDA tensors"
        # every buffer must hold exactly n rays: a short one would be read / written out of bounds by the kernels
        sizes = {"ray_o": (ray_o, 3 * n), "ray_d": (ray_d, 3 * n), "max_dis": (max_dis, n), "work_mask": (work_mask, n),
                 "out.convergent_mask": (out["convergent_mask"], n), "out.points": (out["points"], 3 * n),
                 "out.sdf": (out["sdf"], n), "out.distance": (out["distance"], n)}
        for name, (t, want) in sizes.items():
            if t.numel() != want:
                raise ValueError(f"forward_static: {name} has {t.numel()} elements, expected {want} for {n} rays")
        if status is not None and not (status.is_cuda and status.dtype == torch.int32 and status.numel() >= 1):
            raise ValueError("forward_static: status must be an int32 CUDA tensor")
        with torch.cuda.device(ray_o.device):
            _lib.check(_lib.lib().pir_trace_static(
                _lib.ptr(net.packed().buf), n, _lib.ptr(ray_o), _lib.ptr(ray_d), _lib.ptr(min_dis), _lib.ptr(max_dis),
                _lib.ptr(work_mask), float(self.sdf_threshold), int(self.sphere_tracing_iters), int(self.n_steps),
                int(chunk) if chunk else 0, int(self.max_bisect), _lib.ptr(out["convergent_mask"]),
                _lib.ptr(out["points"]), _lib.ptr(out["sdf"]), _lib.ptr(out["distance"]), _lib.ptr(status),
                _lib.stream_ptr()), "pir_trace_static")
        return out

    def check_status(self, device=None, reset=True):
        """Synchronising check, meant once per training step / rendered image (not per trace): raises if any native
        trace since the last check ran out of bisection rounds, i.e. returned brackets wider than 2*sdf_threshold
        where the reference would have kept halving."""
        bad = []
        want = None if device is None else torch.device(device).index
        for (idx, _stream), acc in self._status.items():
            if want not in (None, idx):
                continue
            torch.cuda.synchronize(idx)       # a trace still running on ANOTHER stream must have landed its OR
            if int(acc.item()) != 0:
                bad.append(idx)
            if reset:
                acc.zero_()
        if bad:
            raise RuntimeError(
                f"RayTracer: the bisection bound (max_bisect={self.max_bisect}) was hit on device(s) {bad}: some hit "
                "points are not converged to 2*sdf_threshold.  Raise RayTracer.max_bisect (<= 64) and trace again.")

    def trace_status(self, device):
        """Synchronising check that the last native trace on `device` did not hit the bisection bound."""
        ws = _trace_ws.get(_lib.stream_key(device))
        if ws is None:
            return 0
        st = ctypes.c_int(0)
        with torch.cuda.device(device):
            _lib.check(_lib.lib().pir_trace_status(_lib.ptr(ws), ctypes.byref(st), _lib.stream_ptr()), "pir_trace_status")
        return st.value

    @torch.no_grad()
    def forward(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask, chunk=None):
        net = getattr(sdf, "sdf_network", None)
        if net is not None and ray_o.is_cuda and net.hip_supported():
            return self._trace_hip(net, ray_o, ray_d, min_dis, max_dis, work_mask, chunk)
        return self._trace_torch(sdf, ray_o, ray_d, min_dis, max_dis, work_mask)

    # ---------------------------------------------------------------- generic path for opaque callables / CPU
    def _trace_torch(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask):
        conv, unf, pts, val, acc = self.sphere_tracing(sdf, ray_o, ray_d, min_dis, max_dis, work_mask)
        if unf.any():
            outside = val[unf] > 0.0
            lo = torch.where(outside, acc[unf], min_dis[unf])
            hi = torch.where(outside, max_dis[unf], acc[unf])
            found, p, s, d = self.ray_sampler(sdf, ray_o[unf], ray_d[unf], lo, hi)
            conv[unf], pts[unf], val[unf], acc[unf] = found, p, s, d
        return {"convergent_mask": conv, "points": pts, "sdf": val, "distance": acc}

    def sphere_tracing(self, sdf, ray_o, ray_d, min_dis, max_dis, work_mask):
        """raytracer.py:126-170."""
        active = work_mask.clone()
        acc = min_dis.clone()
        pts = ray_o + ray_d * acc.unsqueeze(-1)
        val = sdf(pts)
        for it in range(self.sphere_tracing_iters + 1):
            active = active & (val.abs() > self.sdf_threshold) & (acc < max_dis)
            if it == self.sphere_tracing_iters or not active.any():
                break
            step = val[active]
            acc[active] = acc[active] + step
            pts[active] = pts[active] + ray_d[active] * step.unsqueeze(-1)
            val[active] = sdf(pts[active])
        conv = work_mask & ~active & (val.abs() <= self.sdf_threshold) & (acc < max_dis)
        return conv, active, pts, val, acc

    def ray_sampler(self, sdf, ray_o, ray_d, min_dis, max_dis):
        """raytracer.py:172-227."""
        n = self.n_steps
        t = torch.linspace(0, 1, steps=n, device=min_dis.device).float().view(1, n)
        z = min_dis.unsqueeze(-1) + t * (max_dis - min_dis).unsqueeze(-1)
        p = (ray_o.unsqueeze(-2) + ray_d.unsqueeze(-2) * z.unsqueeze(-1)).reshape(-1, 3)
        vals = torch.cat([sdf(c) for c in torch.split(p, self.max_num_pts, dim=0)], dim=0).reshape(-1, n)
        neg = vals < 0
        first = torch.where(neg.any(-1), neg.float().argmax(-1), torch.full_like(neg[:, 0], n, dtype=torch.long))
        found = (first >= 1) & (first < n)
        out_p, out_s, out_d = torch.zeros_like(ray_d), torch.zeros_like(min_dis), torch.zeros_like(min_dis)
        if found.any():
            i = first[found].unsqueeze(-1)
            zf, vf = z[found], vals[found]
            pp, dd, ss = self.rootfind(sdf, vf.gather(-1, i - 1).squeeze(-1), vf.gather(-1, i).squeeze(-1),
                                       zf.gather(-1, i - 1).squeeze(-1), zf.gather(-1, i).squeeze(-1),
                                       ray_o[found], ray_d[found])
            out_p[found], out_s[found], out_d[found] = pp, ss, dd
        return found, out_p, out_s, out_d

    def rootfind(self, sdf, f_low, f_high, d_low, d_high, ray_o, ray_d):
        """raytracer.py:229-250 (every bracket keeps halving while any ray of the call still works)."""
        work = (f_low > 0) & (f_high < 0)
        d_mid = (d_low + d_high) / 2.0
        while work.any():
            f_mid = sdf(ray_o + ray_d * d_mid.unsqueeze(-1))
            pos = f_mid > 0
            d_low, f_low = torch.where(pos, d_mid, d_low), torch.where(pos, f_mid, f_low)
            d_high, f_high = torch.where(pos, d_high, d_mid), torch.where(pos, f_high, f_mid)
            d_mid = (d_low + d_high) / 2.0
            work = work & ((d_high - d_low) > 2 * self.sdf_threshold)
        p_mid = ray_o + ray_d * d_mid.unsqueeze(-1)
        return p_mid, d_mid, sdf(p_mid)


@torch.no_grad()
def intersect_sphere(ray_o, ray_d, r):
    """raytracer.py:253-273: (hit mask, entry distance clamped at 0, exit distance)."""
    d1 = -torch.sum(ray_d * ray_o, dim=-1) / torch.sum(ray_d * ray_d, dim=-1)
    p = ray_o + d1.unsqueeze(-1) * ray_d
    tmp = r * r - torch.sum(p * p, dim=-1)
    d2 = torch.sqrt(torch.clamp(tmp, min=0.0)) / torch.norm(ray_d, dim=-1)
    return tmp > 0.0, torch.clamp(d1 - d2, min=0.0), d1 + d2


class Camera(object):
    """raytracer.py:276-402 (cv2-based `resize` of images is outside the hot path and not provided)."""

    def __init__(self, W, H, K, W2C):
        self.W, self.H = W, H
        self.K, self.W2C = K, W2C
        self.K_inv = torch.inverse(K)
        self.C2W = torch.inverse(W2C)
        self.device = self.K.device
        self._cam21 = None

    def get_fx(self):
        return self.K[0, 0]

    def get_image_size(self):
        return self.H, self.W

    def get_light_pos(self, offset):
        return torch.matmul(offset, self.C2W[:3, :3].transpose(1, 0)) + self.C2W[:3, 3].unsqueeze(0)

    def cam21(self):
        if self._cam21 is None:
            self._cam21 = torch.cat([self.K_inv[:3, :3].reshape(-1), self.C2W[:3, :3].reshape(-1),
                                     self.C2W[:3, 3].reshape(-1)]).contiguous().float()
        return self._cam21

    def get_rays(self, uv):
        dots_sh = list(uv.shape[:-1])
        uv1 = torch.cat((uv.reshape(-1, 2), torch.ones_like(uv.reshape(-1, 2)[..., 0:1])), dim=-1)
        ray_d = torch.matmul(torch.matmul(uv1, self.K_inv[:3, :3].transpose(1, 0)),
                             self.C2W[:3, :3].transpose(1, 0)).reshape(dots_sh + [3])
        ray_d_norm = ray_d.norm(dim=-1)
        ray_d = ray_d / ray_d_norm.unsqueeze(-1)
        ray_o = self.C2W[:3, 3].unsqueeze(0).expand(uv1.shape[0], -1).reshape(dots_sh + [3])
        return ray_o, ray_d, ray_d_norm

    def get_camera_origin(self, prefix_shape=None):
        ray_o = self.C2W[:3, 3]
        if prefix_shape is not None:
            prefix_shape = list(prefix_shape)
            ray_o = ray_o.view([1] * len(prefix_shape) + [3]).expand(prefix_shape + [3])
        return ray_o

    def get_uv(self):
        v, u = torch.meshgrid(torch.arange(self.H, dtype=torch.float32, device=self.device),
                              torch.arange(self.W, dtype=torch.float32, device=self.device), indexing="ij")
        return torch.stack((u, v), dim=-1) + 0.5

    def project(self, points):
        dots_sh = list(points.shape[:-1])
        p = points.reshape(-1, 3)
        p = torch.cat([p, torch.ones_like(p[:, :1])], dim=1)
        uv = torch.matmul(torch.matmul(p, self.W2C.transpose(1, 0)), self.K.transpose(1, 0))
        return (uv[:, :2] / uv[:, 2:3]).view(dots_sh + [2])

    def crop_region(self, trgt_W, trgt_H, center_crop=False, ul_corner=None, images=None):
        K = self.K.clone()
        if ul_corner is not None:
            ul_col, ul_row = ul_corner
        elif center_crop:
            ul_col, ul_row = self.W // 2 - trgt_W // 2, self.H // 2 - trgt_H // 2
        else:
            ul_col = np.random.randint(0, self.W - trgt_W)
            ul_row = np.random.randint(0, self.H - trgt_H)
        K[0, 2] -= ul_col
        K[1, 2] -= ul_row
        cropped = {}
        if images is not None:
            for key, img in images.items():
                assert img.shape[0] == self.H and img.shape[1] == self.W, f"image size for {key} does not match"
                cropped[key] = img[ul_row:ul_row + trgt_H, ul_col:ul_col + trgt_W]
        return Camera(trgt_W, trgt_H, K, self.W2C.clone()), cropped

    def resize(self, factor, image=None):
        """raytracer.py:385-402: a camera with both image sides scaled by factor < 1 (intrinsics rows scaled
        accordingly) and, optionally, the image area-averaged to that size."""
        assert factor < 1.0
        trgt_H, trgt_W = int(self.H * factor), int(self.W * factor)
        K = self.K.clone()
        K[0, :3] *= trgt_W / self.W
        K[1, :3] *= trgt_H / self.H
        camera = Camera(trgt_W, trgt_H, K, self.W2C.clone())
        if image is not None:
            image = _resize_area(image, trgt_W, trgt_H).to(image.device)
        return camera, image


def _area_weights(src, dst, device):
    """[dst, src] weights of an area-averaging downscale (OpenCV's INTER_AREA table, computeResizeAreaTab: the share
    of every source cell that a destination cell of width src/dst covers, normalised by the cell width)."""
    scale = src / dst
    w = torch.zeros(dst, src, dtype=torch.float64)
    for d in range(dst):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, src - f1)
        s1, s2 = int(np.ceil(f1)), min(int(np.floor(f2)), src - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            w[d, s1 - 1] = (s1 - f1) / cell
        w[d, s1:s2] = 1.0 / cell
        if f2 - s2 > 1e-3:
            w[d, s2] = min(min(f2 - s2, 1.0), cell) / cell
    return w.float().to(device)


def _resize_area(image, trgt_W, trgt_H):
    """cv2.resize(image, (trgt_W, trgt_H), interpolation=cv2.INTER_AREA) for an [H, W] or [H, W, C] float image
    without cv2 (not installed here: parity with OpenCV is unpinned; for integer factors this is the plain box mean)."""
    H, W = image.shape[0], image.shape[1]
    img = image.detach().float()
    ry, rx = _area_weights(H, trgt_H, img.device), _area_weights(W, trgt_W, img.device)
    out = torch.einsum("ih,hw...->iw...", ry, img)
    return torch.einsum("jw,iw...->ij...", rx, out).contiguous()


def _camera_rays_hip(camera, uv, is_pixel_grid):
    dev = camera.device
    n = uv.numel() // 2
    ray_o = torch.empty(n, 3, dtype=torch.float32, device=dev)
    ray_d = torch.empty(n, 3, dtype=torch.float32, device=dev)
    ray_n = torch.empty(n, dtype=torch.float32, device=dev)
    hit = torch.empty(n, dtype=torch.uint8, device=dev)
    lo = torch.empty(n, dtype=torch.float32, device=dev)
    hi = torch.empty(n, dtype=torch.float32, device=dev)
    uvp = None if is_pixel_grid else uv.detach().reshape(-1, 2).contiguous().float()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pir_camera_rays(n, int(camera.W), _lib.ptr(uvp), _lib.ptr(camera.cam21()), 1.0,
                                              _lib.ptr(ray_o), _lib.ptr(ray_d), _lib.ptr(ray_n), _lib.ptr(hit),
                                              _lib.ptr(lo), _lib.ptr(hi), _lib.stream_ptr()), "pir_camera_rays")
    return ray_o, ray_d, ray_n, hit.bool(), lo, hi


@torch.no_grad()
def raytrace_pixels(sdf_network, raytracer, uv, camera, mask=None, max_num_rays=200000, _pixel_grid=False):
    """raytracer.py:405-484 -> dict(convergent_mask, points, sdf, distance, depth, uv, ray_o, ray_d, ray_d_norm)."""
    dots_sh = list(uv.shape[:-1])
    native = uv.is_cuda and hasattr(sdf_network, "hip_supported") and sdf_network.hip_supported()
    if native:
        ray_o, ray_d, ray_n, hit, lo, hi = _camera_rays_hip(camera, uv, _pixel_grid)
        work = hit if mask is None else hit & mask.reshape(-1)
        res = raytracer(SdfCallable(sdf_network), ray_o, ray_d, lo, hi, work, chunk=max_num_rays)
    else:
        ray_o, ray_d, ray_n = camera.get_rays(uv)
        ray_o, ray_d, ray_n = ray_o.reshape(-1, 3), ray_d.reshape(-1, 3), ray_n.reshape(-1)
        m = torch.ones_like(ray_n, dtype=torch.bool) if mask is None else mask.reshape(-1)
        sdf = lambda x: sdf_network(x)[..., 0]  # noqa: E731
        parts = []
        for s in range(0, ray_n.shape[0], max_num_rays):
            sl = slice(s, s + max_num_rays)
            hit, lo, hi = intersect_sphere(ray_o[sl], ray_d[sl], r=1.0)
            parts.append(raytracer(sdf, ray_o[sl], ray_d[sl], lo, hi, hit & m[sl]))
        res = {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
    res["depth"] = res["distance"] / ray_n
    out = {}
    for k, v in res.items():
        out[k] = v.reshape(dots_sh + [3]) if v.dim() == 2 else v.reshape(dots_sh)
    out.update({"uv": uv, "ray_o": ray_o.reshape(dots_sh + [3]), "ray_d": ray_d.reshape(dots_sh + [3]),
                "ray_d_norm": ray_n.reshape(dots_sh)})
    return out


def _closing3x3(depth):
    """3x3 morphological closing with a flat structuring element (dilate, then erode), borders ignored."""
    x = depth[None, None]
    x = F.max_pool2d(x, 3, stride=1, padding=1)
    x = -F.max_pool2d(-x, 3, stride=1, padding=1)
    return x[0, 0]


def _fill_holes(results):
    """raytracer.py:630-640: pixels the tracer missed but whose 3x3 closing of the depth map is > 1e-2 take the
    closed depth (and the hit point it implies) and join the hit mask.  In place; no host synchronisation (the
    reference's `.any()` guard only skips work that changes nothing)."""
    depth = _closing3x3(results["depth"])
    new_mask = depth > 1e-2
    update = new_mask & ~results["convergent_mask"]
    any_update = update.any()             # 0-dim device tensor: selects below without a host read
    results["depth"] = torch.where(update, depth, results["depth"])
    results["convergent_mask"] = torch.where(any_update, new_mask, results["convergent_mask"])
    # as in the reference, distance and points of EVERY pixel are then re-derived from the depth map
    distance = results["depth"] * results["ray_d_norm"]
    results["distance"] = torch.where(any_update, distance, results["distance"])
    results["points"] = torch.where(any_update, results["ray_o"] + results["ray_d"] * distance.unsqueeze(-1),
                                    results["points"])


_SOBEL = {}


def _sobel_magnitude(depth):
    """kornia.filters.sobel without kornia: 3x3 Sobel kernels normalised by 8, replicate padding,
    sqrt(gx^2 + gy^2 + 1e-6) (kornia's published algorithm; parity with kornia itself unpinned)."""
    w = _SOBEL.get(depth.device)
    if w is None:      # built once per device: a host-to-device copy per call waits for everything queued on the stream
        kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]]) / 8.0
        w = _SOBEL[depth.device] = torch.stack([kx, kx.t()])[:, None].to(depth.device)
    x = F.pad(depth[None, None], (1, 1, 1, 1), mode="replicate")
    g = F.conv2d(x, w)[0]
    return torch.sqrt(g[0] * g[0] + g[1] * g[1] + 1e-6)


def unique(x, dim=-1):
    """Sorted unique values of x and the index of the first occurrence of each (raytracer.py:487-494).  The
    reference scatters a flipped index range and relies on the last write winning, which only holds on the CPU;
    an `amin` scatter-reduce gives the same answer deterministically on the GPU."""
    vals, inverse = torch.unique(x, return_inverse=True, dim=dim)
    perm = torch.arange(inverse.size(dim), dtype=inverse.dtype, device=inverse.device)
    first = torch.full((vals.size(dim),), inverse.size(dim), dtype=inverse.dtype, device=inverse.device)
    return vals, first.scatter_reduce_(0, inverse, perm, reduce="amin")


def _edge_walk_torch(pts, cam_o, sdf_network, max_step, step_size, dot_threshold, max_num_rays):
    """The walk with eager ops (CPU tensors, SDF architectures the kernels do not cover); updates pts in place."""
    dev = pts.device
    found = torch.zeros(pts.shape[0], dtype=torch.bool, device=dev)
    for s in range(0, pts.shape[0], max_num_rays):       # the reference walks max_num_rays points at a time
        cur, cur_found = pts[s:s + max_num_rays], found[s:s + max_num_rays]
        todo = torch.arange(cur.shape[0], device=dev)    # indices of the points still walking
        for it in range(max_step + 1):
            x = cur[todo]
            view = cam_o - x
            view = view / (view.norm(dim=-1, keepdim=True) + 1e-10)
            sdf, _, nrm = sdf_network.get_all(x.clone(), is_training=False)
            nrm = nrm / (nrm.norm(dim=-1, keepdim=True) + 1e-10)
            dot = (nrm * view).sum(dim=-1)
            walking = dot.abs() > dot_threshold
            cur_found[todo[~walking]] = True
            if it >= max_step or not bool(walking.any()):
                break
            step = nrm - view / dot.unsqueeze(-1)
            step = step / (step.norm(dim=-1, keepdim=True) + 1e-10)
            step = step - sdf.reshape(-1, 1) * nrm       # stay near the zero level set
            todo = todo[walking]
            cur[todo] += (step_size * step)[walking]
    return found


@torch.no_grad()
def locate_edge_points(camera, walk_start_points, sdf_network, max_step, step_size, dot_threshold,
                       max_num_rays=200000, mask=None):
    """Walk on the surface from the depth-edge pixels until the normal is perpendicular to the view direction
    (raytracer.py:497-615) -> dict(edge_mask [H,W], edge_points [E,3], edge_uv [E,2], edge_pixel_idx [E]).
    On the GPU the whole walk of all points is ONE launch (pir_edge_walk)."""
    if mask is None:
        mask = torch.ones_like(walk_start_points[..., 0]).bool()
    dev = walk_start_points.device
    cam_o = camera.get_camera_origin()
    pts = walk_start_points[mask].reshape(-1, 3).detach().clone()
    if pts.is_cuda and FUSED_EDGE_WALK and getattr(sdf_network, "hip_supported", lambda: False)():
        # the whole walk in one launch (pir_edge_walk); points evolve independently, so the reference's max_num_rays
        # batching has no effect on the result
        from .. import sdf_hip
        pts, found = sdf_hip.edge_walk(sdf_network.packed(), pts, cam_o, max_step, step_size, dot_threshold)
    else:
        found = _edge_walk_torch(pts, cam_o, sdf_network, max_step, step_size, dot_threshold, max_num_rays)
    edge_points = pts[found]
    edge_mask = torch.zeros(camera.H, camera.W, dtype=torch.bool, device=dev)
    edge_uv = torch.zeros_like(edge_points[..., :2])
    update_pixels = torch.zeros(0, dtype=torch.long, device=dev)
    if edge_points.shape[0] > 0:
        edge_uv = camera.project(edge_points)
        pix = torch.floor(edge_uv).long()
        update_pixels = pix[:, 1] * camera.W + pix[:, 0]
        ok = (update_pixels < camera.H * camera.W) & (update_pixels >= 0)
        update_pixels, edge_points, edge_uv = update_pixels[ok], edge_points[ok], edge_uv[ok]
        if update_pixels.shape[0] > 0:
            update_pixels, first = unique(update_pixels, dim=0)   # one edge point per pixel: the first found
            edge_points, edge_uv = edge_points[first], edge_uv[first]
            edge_mask.view(-1)[update_pixels] = True
    return {"edge_mask": edge_mask, "edge_points": edge_points, "edge_uv": edge_uv, "edge_pixel_idx": update_pixels}


@torch.no_grad()
def raytrace_camera(camera, sdf_network, raytracer, max_num_rays=200000, fill_holes=False, detect_edges=False):
    """raytracer.py:618-666.  fill_holes / detect_edges use kornia-free 3x3 closing / Sobel filters (parity with
    kornia unpinned)."""
    results = raytrace_pixels(sdf_network, raytracer, camera.get_uv(), camera, max_num_rays=max_num_rays,
                              _pixel_grid=True)
    results["depth"] = results["depth"] * results["convergent_mask"].float()
    if fill_holes:
        _fill_holes(results)
    if detect_edges:
        depth_edge_mask = (_sobel_magnitude(results["depth"]) > 1e-2) & results["convergent_mask"]
        results.update(locate_edge_points(camera, results["points"], sdf_network, max_step=16, step_size=1e-3,
                                          dot_threshold=5e-2, max_num_rays=max_num_rays, mask=depth_edge_mask))
        results["convergent_mask"] = results["convergent_mask"] & ~results["edge_mask"]
    else:
        results["edge_mask"] = torch.zeros_like(results["convergent_mask"])
    return results


def _light_visibility_torch(light_pos, points, sdf_network, inv_s, n_samples, is_training):
    """The march written with PyTorch ops: CPU host-logic tests, networks the kernels do not cover, and an inv_s
    that needs a gradient (the reference always passes a detached one, render_surface.py:241)."""
    device = points.device
    bs = points.shape[0]
    delta = points - light_pos
    length = torch.norm(delta, dim=-1, keepdim=True)
    rays_d = delta / length
    discard = 10
    z = torch.linspace(0, 1, n_samples, device=device)[:-discard]
    n = n_samples - discard
    z = z.view(1, n) * length
    dists = torch.cat([z[..., 1:] - z[..., :-1], torch.full((bs, 1), 1.0 / n, device=device)], dim=-1)
    mid = z + dists * 0.5
    pts = (light_pos + rays_d[..., None, :] * mid[..., :, None]).reshape(-1, 3)
    dirs = rays_d[..., None, :].expand(bs, n, 3).reshape(-1, 3)
    if hasattr(sdf_network, "get_sdf_normal"):
        sdf, grad = sdf_network.get_sdf_normal(pts, is_training=is_training)
        sdf = sdf.reshape(-1, 1)
    else:
        sdf, _, grad = sdf_network.get_all(pts, is_training=is_training)
    cos = -F.relu(-(dirs * grad).sum(-1, keepdim=True))
    half = cos * dists.reshape(-1, 1) * 0.5
    k = inv_s.reshape(1, 1)
    prev_cdf = torch.sigmoid((sdf - half) * k)
    next_cdf = torch.sigmoid((sdf + half) * k)
    alpha = ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).reshape(bs, n).clip(0.0, 1.0)
    trans = torch.cumprod(torch.cat([torch.ones(bs, 1, device=device), 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    acc_map = (1.0 - (alpha * trans).sum(dim=-1, keepdim=True)).reshape(bs)
    return acc_map if is_training else acc_map.clone().detach()


def light_visibility(light_pos, points, sdf_network, inv_s, n_samples=128, is_training=True):
    """Transmittance from the light to each surface point by a NeuS-style march (raytracer.py:669-729).

    On the GPU: sample positions (pir_visibility_points), value + normal of the n_samples-10 samples per point
    (pir_sdf_get_all), opacity compositing (pir_visibility_forward); their backward kernels under autograd."""
    inv_s = torch.as_tensor(inv_s, dtype=torch.float32, device=points.device)
    native = (points.is_cuda and getattr(sdf_network, "hip_supported", lambda: False)() and not inv_s.requires_grad
              and 2 <= n_samples - 10 <= 512)
    if not native:
        return _light_visibility_torch(light_pos, points, sdf_network, inv_s, n_samples, is_training)
    from .. import visibility_hip
    with torch.set_grad_enabled(is_training and torch.is_grad_enabled()):
        delta = points - light_pos
        length = torch.norm(delta, dim=-1, keepdim=True)
        rays_d = delta / length
        pts = visibility_hip.march_points(light_pos, rays_d, length, n_samples)
        sdf, grad = sdf_network.get_sdf_normal(pts, is_training=is_training)
        vis = visibility_hip.composite(sdf, grad, rays_d, length, inv_s, n_samples)
    return vis if is_training else vis.detach()


def render_light_visibility(results, camera, light_pos, sdf_network, inv_s, n_samples=128, max_num_pts=2500,
                            is_training=True):
    """raytracer.py:732-777: adds results["vis"] [H,W] (0 outside the hit/edge mask)."""
    dots_sh = list(results["convergent_mask"].shape)
    mask = (results["convergent_mask"] | results["edge_mask"]).reshape(-1)
    vis = torch.zeros(mask.shape, dtype=torch.float32, device=mask.device)
    idx = mask_indices(mask)                 # one nonzero (one host sync) for the gather and the scatter
    if idx.numel() > 0:
        pts = results["points"].reshape(-1, 3).index_select(0, idx)
        if pts.is_cuda:      # the HIP march walks the points in chunks sized to its own scratch
            v = light_visibility(light_pos, pts, sdf_network, inv_s, n_samples=n_samples, is_training=is_training)
        else:                # eager expressions: max_num_pts points at a time, as the reference (raytracer.py:745-760)
            v = torch.cat([light_visibility(light_pos, p, sdf_network, inv_s, n_samples=n_samples, is_training=is_training)
                           for p in torch.split(pts, max(1, int(max_num_pts)), dim=0)], dim=0)
        with torch.set_grad_enabled(is_training):
            vis = vis.index_put((idx,), v)
    results.update({"vis": vis.reshape(dots_sh)})


def render_normal_and_color(results, sdf_network, color_network_dict, dino_network, render_fn, is_training=False,
                            max_num_pts=320000, light_pos=None, roughness_warmup=False):
    """raytracer.py:780-865: shades the convergent pixels and merges render_fn's dict into `results`."""
    dots_sh = list(results["convergent_mask"].shape)
    mask = results["convergent_mask"].reshape(-1)
    empty = torch.zeros(0, dtype=torch.float32, device=mask.device)
    fused = False
    idx = mask_indices(mask)                 # shared with render_fn's scatter: one nonzero / host sync per call
    any_hit = idx.numel() > 0
    if any_hit:
        points = results["points"].reshape(-1, 3).index_select(0, idx)
        ray_d = results["ray_d"].reshape(-1, 3).index_select(0, idx)
        ray_o = results["ray_o"].reshape(-1, 3).index_select(0, idx)
        sdf, feature, normal = sdf_network.get_all(points, is_training=is_training)
        dino_feature = dino_network(points, is_training=False)
        fused = points.is_cuda and FUSED_SHADE_INPUTS and getattr(render_fn, "accepts_unit_normals", False)
        if fused:
            # reparam_points + normal normalisation + lightdir / distance in one kernel each way (pir_shade_inputs_*)
            from .. import shade_hip
            with torch.set_grad_enabled(is_training):
                points, normal, lightdir, distance = shade_hip.shade_inputs(points, sdf, normal, ray_d, ray_o,
                                                                            light_pos, is_training)
        elif is_training:
            points = reparam_points(points, normal.detach(), -ray_d.detach(), sdf)
    else:
        points = ray_d = ray_o = normal = feature = dino_feature = empty
    with torch.set_grad_enabled(is_training):
        if not fused:
            lightdir = distance = None
            if light_pos is not None:
                if any_hit:
                    lightdir = F.normalize(points - light_pos, eps=1e-10)
                    distance = (points - light_pos).norm(dim=-1, keepdim=True)
                else:
                    lightdir = empty
        render_results = render_fn(mask, color_network_dict, ray_o, ray_d, points, normal, feature,
                                   dino_f=dino_feature, is_training=is_training, lightdir=lightdir, distance=distance,
                                   roughness_warmup=roughness_warmup)
    for k, v in render_results.items():
        v = v.reshape(dots_sh + [-1])
        results[k] = v.squeeze(-1) if v.shape[-1] == 1 else v


_nz_cache = [None, None, None]     # [weakref to the mask tensor, its version, indices]


def mask_indices(mask):
    """mask.nonzero() of a flat bool mask, remembered for the SAME tensor object at the same version: the drivers
    below and render_fn index several arrays by one mask; a boolean index costs a nonzero + a host sync each time."""
    import weakref
    ref, ver, idx = _nz_cache
    if ref is not None and ref() is mask and ver == mask._version:
        return idx
    idx = mask.nonzero(as_tuple=True)[0]
    _nz_cache[:] = [weakref.ref(mask), mask._version, idx]
    return idx


def _put_rows(img, idx, rows):
    """img.view(-1, C)[idx] = rows, out of place (keeps autograd history of both operands)."""
    shp = img.shape
    flat = img.reshape(shp[0] * shp[1], -1)
    return flat.index_put((idx,), rows.reshape(idx.shape[0], -1)).reshape(shp)


def render_edge_pixels(results, camera, sdf_network, raytracer, color_network_dict, dino_network, render_fn,
                       is_training=False, light_pos=None, roughness_warmup=False):
    """Colour of an edge pixel = area-weighted mix of the colours traced either side of the silhouette, with the
    weight differentiable w.r.t. the edge point (raytracer.py:872-951)."""
    edge_points, edge_uv, edge_pixel_idx = results["edge_points"], results["edge_uv"], results["edge_pixel_idx"]
    centre = torch.floor(edge_uv) + 0.5
    edge_sdf, _, edge_grads = sdf_network.get_all(edge_points, is_training=is_training)
    g = edge_grads.detach()
    edge_normals = g / (g.norm(dim=-1, keepdim=True) + 1e-10)
    if is_training:
        edge_points = reparam_points(edge_points, g, edge_normals, edge_sdf)
        edge_uv = camera.project(edge_points)
    n2d = torch.matmul(edge_normals, camera.W2C[:3, :3].transpose(1, 0))[:, :2]
    n2d = n2d / (n2d.norm(dim=-1, keepdim=True) + 1e-10)
    pixel_radius = 0.707
    dot2d = torch.sum((edge_uv - centre) * n2d, dim=-1)
    alpha = 2 * torch.arccos(torch.clamp(dot2d / pixel_radius, min=0.0, max=1.0))
    w_pos = (1.0 - (alpha - torch.sin(alpha)) / (2.0 * np.pi)).unsqueeze(-1)

    side_uvs = (centre - pixel_radius * n2d, centre + pixel_radius * n2d)        # positive side, negative side
    E = edge_uv.shape[0]
    if 0 < E <= 200000:
        # Both sides in ONE trace and ONE shading call (half the launches and gradient accumulations).  Rays are
        # independent except for the bisection loop, which the reference runs until every ray of a CALL is done
        # (raytracer.py:234): max_num_rays = E keeps the two sides in separate chunks of the tracer, so each side
        # bisects exactly as in a call of its own.
        r = raytrace_pixels(sdf_network, raytracer, torch.cat(side_uvs, dim=0), camera, max_num_rays=E)
        render_normal_and_color(r, sdf_network, color_network_dict, dino_network, render_fn,
                                is_training=is_training, light_pos=light_pos, roughness_warmup=roughness_warmup)
        pos = {k: (v[:E] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 * E else v) for k, v in r.items()}
        neg = {k: (v[E:] if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == 2 * E else v) for k, v in r.items()}
    else:
        sides = []
        for side_uv in side_uvs:
            r = raytrace_pixels(sdf_network, raytracer, side_uv, camera)
            render_normal_and_color(r, sdf_network, color_network_dict, dino_network, render_fn,
                                    is_training=is_training, light_pos=light_pos, roughness_warmup=roughness_warmup)
            sides.append(r)
        pos, neg = sides

    def mix(key):
        a, b = pos[key], neg[key]
        w = w_pos if a.dim() == 2 else w_pos.squeeze(-1)
        return a * w + b * (1.0 - w)

    results["color"] = _put_rows(results["color"], edge_pixel_idx, mix("color"))
    results["normal"] = _put_rows(results["normal"], edge_pixel_idx, edge_grads)
    for key in ("diffuse_albedo", "specular_albedo", "specular_roughness", "dino_feature"):   # visualisation maps
        results[key] = _put_rows(results[key], edge_pixel_idx, mix(key))
    results["edge_pos_neg_normal"] = torch.cat([pos["normal"][pos["convergent_mask"]],
                                                neg["normal"][neg["convergent_mask"]]], dim=0)
    results["uv"] = _put_rows(results["uv"], edge_pixel_idx, edge_uv.detach())
    results["points"] = _put_rows(results["points"], edge_pixel_idx, edge_points.detach())


def render_indirect_lighting(results, sdf_network, color_network_dict, dino_network, raytracer, render_fn,
                             light_pos=None, roughness_warmup=False, is_training=False, sec_is_training=False,
                             diff_geo=True):
    """One mirror bounce: trace the reflected ray of every hit, shade the secondary hit under the point light,
    then shade the primary point lit by that radiance; a BlendNetwork weight per pixel
    (raytracer.py:1003-1189).  Adds results["indirect_color"] [H,W,3] and ["indirect_blend_coef"] [H,W,1]."""
    dots_sh = list(results["convergent_mask"].shape)
    device = results["points"].device
    points_mask = results["convergent_mask"].reshape(-1)
    n_pix = points_mask.shape[0]
    indirect_color = torch.zeros(n_pix, 3, device=device)
    indirect_blend_coef = torch.ones(n_pix, device=device)
    if points_mask.any():
        all_points = results["points"].reshape(-1, 3)[points_mask].clone()
        all_ray_o = results["ray_o"].reshape(-1, 3)[points_mask]
        all_ray_d = results["ray_d"].reshape(-1, 3)[points_mask]
        colors, coefs = [], []
        sdf_fn = SdfCallable(sdf_network)
        for points, ray_o, ray_d in zip(torch.split(all_points, 200000), torch.split(all_ray_o, 200000),
                                        torch.split(all_ray_d, 200000)):
            with torch.no_grad():
                normals = sdf_network.get_all(points.clone(), is_training=False)[2]
                normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
                sample_dirs = torch.sum(-ray_d * normals, dim=-1, keepdim=True) * normals * 2 + ray_d
                sample_dirs = sample_dirs / (sample_dirs.norm(dim=-1, keepdim=True) + 1e-10)
                hit, lo, hi = intersect_sphere(points, sample_dirs, r=1.0)
                sec = raytracer(sdf_fn, points, sample_dirs, lo.clamp(min=0.001, max=3.00),
                                hi.clamp(min=0.001, max=3.00), hit)
                sec_mask, sec_points = sec["convergent_mask"], sec["points"]
                if light_pos is not None:   # un-colocated light: drop secondary points that do not see the camera
                    z = torch.linspace(0.01, 0.8, 20, device=device)
                    dirs = ray_o[sec_mask] - sec_points[sec_mask]
                    dirs = dirs / (dirs.norm(dim=-1, keepdim=True) + 1e-10)
                    probe = sec_points[sec_mask][:, None, :] + dirs[:, None, :] * z[None, :, None]
                    probe_sdf = sdf_fn(probe.reshape(-1, 3)).reshape(-1, 20)
                    keep = torch.zeros_like(sec_mask)
                    keep[sec_mask] = ~((probe_sdf < 0).sum(dim=-1) > 0)
                    sec_mask = keep
            # ---- radiance leaving the secondary point towards the primary one
            hit_points = sec_points[sec_mask]
            hit_ray_o, hit_ray_d = points[sec_mask], sample_dirs[sec_mask]
            lightdir = hit_points - (ray_o[sec_mask] if light_pos is None else light_pos)
            hit_distance = lightdir.norm(dim=-1, keepdim=True)
            lightdir = lightdir / (lightdir.norm(dim=-1, keepdim=True) + 1e-10)
            hit_sdf, hit_feature, hit_normal = sdf_network.get_all(hit_points, is_training=sec_is_training)
            hit_dino = dino_network(hit_points, is_training=sec_is_training)
            if diff_geo and sec_is_training:
                hit_points = reparam_points(hit_points, hit_normal.detach(), -hit_ray_d.detach(), hit_sdf)
            with torch.set_grad_enabled(sec_is_training):
                sec_results = render_fn(sec_mask, color_network_dict, hit_ray_o, hit_ray_d, hit_points, hit_normal,
                                        hit_feature, dino_f=hit_dino, light=None, lightdir=lightdir,
                                        distance=hit_distance, roughness_warmup=roughness_warmup,
                                        is_training=sec_is_training, EDirectRadiance=True)
            trace_radiance = sec_results["color"][sec_mask]
            # ---- primary point lit from the mirror direction by that radiance
            p_ray_o, p_ray_d, p_points, p_dirs = ray_o[sec_mask], ray_d[sec_mask], points[sec_mask], sample_dirs[sec_mask]
            p_sdf, p_feature, p_normal = sdf_network.get_all(p_points, is_training=is_training)
            p_dino = dino_network(p_points, is_training=False)
            if diff_geo and is_training:
                p_points = reparam_points(p_points, p_normal.detach(), -p_ray_d.detach(), p_sdf)
            with torch.set_grad_enabled(is_training):
                prim = render_fn(sec_mask, color_network_dict, p_ray_o, p_ray_d, p_points, p_normal, p_feature,
                                 p_dino, light=trace_radiance, lightdir=-p_dirs, distance=None,
                                 roughness_warmup=roughness_warmup, is_training=is_training, EDirectRadiance=False)
                blend = torch.ones(sec_mask.shape[0], dtype=torch.float32, device=device)
                if sec_mask.any():
                    roughness = prim["specular_roughness"][sec_mask].reshape(-1, 1).detach()
                    n1 = p_normal.detach()
                    n1 = n1 / (n1.norm(dim=-1, keepdim=True) + 1e-10)
                    dot_n_wo = torch.sum(p_dirs * n1, dim=-1, keepdim=True)
                    distance = torch.norm(p_points - hit_points, dim=-1, keepdim=True)
                    b = color_network_dict["blend_network"](distance, roughness, dot_n_wo).squeeze(-1)
                    blend = blend.index_put((sec_mask.nonzero(as_tuple=True)[0],), b)
            colors.append(prim["color"])
            coefs.append(blend)
        idx = points_mask.nonzero(as_tuple=True)[0]
        indirect_color = indirect_color.index_put((idx,), torch.cat(colors, dim=0))
        indirect_blend_coef = indirect_blend_coef.index_put((idx,), torch.cat(coefs, dim=0))
    results.update({"indirect_color": indirect_color.reshape(dots_sh + [3]),
                    "indirect_blend_coef": indirect_blend_coef.reshape(dots_sh + [1])})


def render_camera(camera, sdf_network, raytracer, color_network_dict, dino_network, render_fn, fill_holes=True,
                  handle_edges=True, inv_s=0.0, render_visibility=False, max_num_pts=512, light_pos=None,
                  roughness_warmup=False, indirect_warmup=False, is_training=False):
    """raytracer.py:1198-1275: trace (+ edge walk), visibility, direct shading, one-bounce indirect term
    (unless indirect_warmup), edge pixels."""
    results = raytrace_camera(camera, sdf_network, raytracer, max_num_rays=200000, fill_holes=fill_holes,
                              detect_edges=handle_edges)
    if render_visibility:
        render_light_visibility(results, camera, light_pos, sdf_network, inv_s, n_samples=128,
                                max_num_pts=max_num_pts, is_training=is_training)
    render_normal_and_color(results, sdf_network, color_network_dict, dino_network, render_fn,
                            is_training=is_training, max_num_pts=320000, light_pos=light_pos,
                            roughness_warmup=roughness_warmup)
    if not indirect_warmup:
        render_indirect_lighting(results, sdf_network, color_network_dict, dino_network, raytracer, render_fn,
                                 roughness_warmup=roughness_warmup, is_training=is_training, diff_geo=True)
    if handle_edges and results["edge_mask"].sum() > 0:
        render_edge_pixels(results, camera, sdf_network, raytracer, color_network_dict, dino_network, render_fn,
                           is_training=is_training, light_pos=light_pos, roughness_warmup=roughness_warmup)
    return results


def render_camera_visibility(camera, sdf_network, raytracer, fill_holes=True, handle_edges=True, is_training=False,
                             inv_s=0.0, render_visibility=False, max_num_pts=4000, light_pos=None):
    """raytracer.py:1278-1310: trace + 256-sample visibility bake."""
    results = raytrace_camera(camera, sdf_network, raytracer, max_num_rays=200000, fill_holes=fill_holes,
                              detect_edges=handle_edges)
    if render_visibility:
        render_light_visibility(results, camera, light_pos, sdf_network, inv_s, n_samples=256,
                                max_num_pts=max_num_pts, is_training=is_training)
    return results


def render_dino(results, dino_network, max_num_pts=320000, is_training=True):
    """DINO field at the hit points, zero elsewhere -> results["dino"] [H, W, 384] (raytracer.py:1313-1352)."""
    dots_sh = list(results["convergent_mask"].shape)
    mask = results["convergent_mask"].reshape(-1)
    dino = torch.zeros(mask.shape[0], dimension, dtype=torch.float32, device=mask.device)
    if mask.any():
        feat = dino_network(results["points"].reshape(-1, 3)[mask], is_training=is_training)
        dino = dino.index_put((mask.nonzero(as_tuple=True)[0],), feat)
    results.update({"dino": dino.reshape(dots_sh + [-1])})


def dino_camera(camera, sdf_network, raytracer, dino_network, is_training=True):
    """Trace the camera and evaluate the surface DINO feature (raytracer.py:1355-1379)."""
    results = raytrace_camera(camera, sdf_network, raytracer, max_num_rays=200000)
    render_dino(results, dino_network, max_num_pts=320000, is_training=is_training)
    return results
