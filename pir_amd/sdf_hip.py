"""Host side of the SDF-network kernels: packed weights, forward, get_all (value / feature / normal)."""
import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib

_SDF_SHAPES = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
_THETA_FLOATS = sum(a * b + a for a, b in _SDF_SHAPES)
_scratch_cache = {}


def supported_architecture(shapes: Sequence[Tuple[int, int]]) -> bool:
    return [tuple(s) for s in shapes] == _SDF_SHAPES


def scratch_for(device, n_workgroups: Optional[int] = None) -> torch.Tensor:
    """Per-(device, stream) scratch for the resident workgroups of the get_all sweeps (2 per CU)."""
    per = _lib.lib().pir_sdf_scratch_bytes_per_workgroup()
    if n_workgroups is None:
        n_workgroups = 2 * torch.cuda.get_device_properties(device).multi_processor_count
    key = _lib.stream_key(device) + (n_workgroups,)
    buf = _scratch_cache.get(key)
    if buf is None:
        buf = torch.empty(per * n_workgroups, dtype=torch.uint8, device=device)
        _scratch_cache[key] = buf
    return buf


class PackedSdf:
    """Weights of one SDFNetwork in MFMA fragment order (built once per optimisation step)."""

    def __init__(self, weights: List[torch.Tensor], biases: List[torch.Tensor]):
        L = _lib.lib()
        assert len(weights) == 9 and len(biases) == 9
        assert supported_architecture([w.shape for w in weights]), "unsupported SDF architecture for the HIP path"
        dev = weights[0].device
        assert dev.type == "cuda"
        self._keep = [w.detach().contiguous().float() for w in weights] + [b.detach().contiguous().float() for b in biases]
        self.device = dev
        self.buf = torch.empty(L.pir_sdf_packed_bytes(), dtype=torch.uint8, device=dev)
        wp = (ctypes.c_void_p * 9)(*[t.data_ptr() for t in self._keep[:9]])
        bp = (ctypes.c_void_p * 9)(*[t.data_ptr() for t in self._keep[9:]])
        with torch.cuda.device(dev):
            _lib.check(L.pir_sdf_pack(wp, bp, _lib.ptr(self.buf), _lib.stream_ptr()), "pir_sdf_pack")

    def forward(self, x: torch.Tensor, want_feature: bool = False):
        x = x.detach().reshape(-1, 3).contiguous().float()
        M = x.shape[0]
        sdf = torch.empty(M, dtype=torch.float32, device=x.device)
        feat = torch.empty(M, 256, dtype=torch.float32, device=x.device) if want_feature else None
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().pir_sdf_forward(_lib.ptr(self.buf), _lib.ptr(x), M, _lib.ptr(sdf), _lib.ptr(feat),
                                                  _lib.stream_ptr()), "pir_sdf_forward")
        return (sdf, feat) if want_feature else sdf

    def get_all(self, x: torch.Tensor, want_feature: bool = True):
        x = x.detach().reshape(-1, 3).contiguous().float()
        M = x.shape[0]
        sdf = torch.empty(M, dtype=torch.float32, device=x.device)
        normal = torch.empty(M, 3, dtype=torch.float32, device=x.device)
        feat = torch.empty(M, 256, dtype=torch.float32, device=x.device) if want_feature else None
        scratch = scratch_for(x.device)
        with torch.cuda.device(x.device):
            _lib.check(_lib.lib().pir_sdf_get_all(_lib.ptr(self.buf), _lib.ptr(x), M, _lib.ptr(sdf), _lib.ptr(feat),
                                                  _lib.ptr(normal), _lib.ptr(scratch), scratch.numel(),
                                                  _lib.stream_ptr()), "pir_sdf_get_all")
        return sdf, feat, normal


_bwd_scratch_cache = {}
BWD_CHUNK_POINTS = 262144   # the C ABI caps a pass at 262 144 points (8.7 GB of scratch); fewer passes = fewer slab reductions
_BWD_MIN_POINTS = 4096


def backward_scratch_for(device, points: int = None, chunk_points: int = None) -> torch.Tensor:
    """Scratch of the SDF backward for a call over `points` points: ONE buffer per (device, stream), sized to the
    largest call seen on that stream (point counts rounded up to a power of two, at most chunk_points = 262 144 points
    = 8.7 GB; a 128 x 128 training patch needs ~0.3 GB).  A buffer that has to grow replaces the old one.
    `free_scratch()` drops every cached buffer (e.g. after a transient stream is gone)."""
    chunk_points = chunk_points or BWD_CHUNK_POINTS
    want = chunk_points if points is None else min(chunk_points, max(_BWD_MIN_POINTS, 1 << max(0, int(points) - 1).bit_length()))
    key = _lib.stream_key(device)
    buf = _bwd_scratch_cache.get(key)
    need = _lib.lib().pir_sdf_backward_scratch_bytes(want)
    if buf is None or buf.numel() < need:
        _bwd_scratch_cache.pop(key, None)
        buf = None                      # release the smaller buffer before allocating the larger one
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _bwd_scratch_cache[key] = buf
    return buf


def free_scratch():
    """Drop every cached scratch / workspace buffer of this module (they are re-created on demand).  Buffers are
    cached per (device, stream): call this after work on short-lived streams, or to hand the memory back."""
    _scratch_cache.clear()
    _bwd_scratch_cache.clear()


def get_all_backward(packed: "PackedSdf", x, g_sdf, g_feat, g_normal, need_gx: bool, chunk_points: int = None):
    """dL/dW_l, dL/db_l (effective weights, lists of 9) and dL/dx for upstream grads of (sdf, feat, normal)."""
    x = x.detach().reshape(-1, 3).contiguous().float()
    M = x.shape[0]
    dev = x.device
    # one zero-filled allocation for all 18 gradient tensors + dL/dx (one fill launch instead of 19)
    sizes = [a * b for a, b in _SDF_SHAPES] + [a for a, _ in _SDF_SHAPES] + [3 * M if need_gx else 0]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    parts = torch.split(flat, sizes)
    gws = [parts[i].view(_SDF_SHAPES[i]) for i in range(9)]
    gbs = list(parts[9:18])
    gx = parts[18].view(M, 3) if need_gx else None
    gws[0]._pir_flat = flat[:_THETA_FLOATS]     # [dW_0 .. dW_8 | db_0 .. db_8], the layout of SDFNetwork.theta()

    def prep(g, shape):
        if g is None:
            return None
        return g.detach().reshape(shape).contiguous().float()

    g_sdf, g_feat, g_normal = prep(g_sdf, (M,)), prep(g_feat, (M, 256)), prep(g_normal, (M, 3))
    scratch = backward_scratch_for(dev, M, chunk_points)
    wp = (ctypes.c_void_p * 9)(*[t.data_ptr() for t in gws])
    bp = (ctypes.c_void_p * 9)(*[t.data_ptr() for t in gbs])
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pir_sdf_get_all_backward(
            _lib.ptr(packed.buf), _lib.ptr(x), M, _lib.ptr(g_sdf), _lib.ptr(g_feat), _lib.ptr(g_normal), _lib.ptr(gx),
            wp, bp, _lib.ptr(scratch), scratch.numel(), _lib.stream_ptr()), "pir_sdf_get_all_backward")
    return gws, gbs, gx


def flat_of(gws, gbs):
    """The flat [dW | db] tensor behind the gradient views get_all_backward returned."""
    return gws[0]._pir_flat


def edge_walk(packed: "PackedSdf", points: torch.Tensor, cam_o: torch.Tensor, max_step: int, step_size: float,
              dot_threshold: float):
    """The surface walk of locate_edge_points (src/renderer/raytracer.py:497-560) in one launch (pir_edge_walk):
    -> (final points [n,3], found [n] bool)."""
    pts = points.detach().reshape(-1, 3).contiguous().float().clone()
    n = pts.shape[0]
    found = torch.zeros(n, dtype=torch.uint8, device=pts.device)
    origin = cam_o.detach().reshape(3).contiguous().float()
    scratch = scratch_for(pts.device)
    with torch.cuda.device(pts.device):
        _lib.check(_lib.lib().pir_edge_walk(_lib.ptr(packed.buf), n, _lib.ptr(pts), _lib.ptr(origin), int(max_step),
                                            float(step_size), float(dot_threshold), _lib.ptr(found), _lib.ptr(scratch),
                                            scratch.numel(), _lib.stream_ptr()), "pir_edge_walk")
    return pts, found.bool()
