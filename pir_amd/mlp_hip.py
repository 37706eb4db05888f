"""Host side of the fused ReLU-MLP kernels (csrc/mlp.hip): the material networks and the DINO feature field
(reference: src/network.py:146-267 RenderingNetwork, :382-467 DINONetwork).

A `PackedMlp` holds one network's effective (weight-normed) matrices in MFMA fragment order plus its bias / head
vectors in ONE device buffer, and the layer plan the kernels read.  It is rebuilt whenever a parameter changed.
"""
import ctypes
import math
from typing import List, Optional

import torch

from . import _lib

_ACT_KB = 32            # k-blocks of a 256-wide activation
_HIDDEN = 256


class PirMlpPlan(ctypes.Structure):
    """include/pir_hip.h: PirMlpPlan"""
    _fields_ = [("kin_kb", ctypes.c_int), ("n_hidden", ctypes.c_int), ("skip", ctypes.c_int), ("n_out", ctypes.c_int),
                ("prod_in", ctypes.c_int), ("prod_h", ctypes.c_int * 8), ("prod_skip_x", ctypes.c_int),
                ("prod_out", ctypes.c_int), ("bias_off", ctypes.c_int), ("head_off", ctypes.c_int),
                ("out_bias_off", ctypes.c_int), ("has_bwd", ctypes.c_int), ("rprod_h", ctypes.c_int * 8),
                ("rprod_in", ctypes.c_int), ("rprod_skip_x", ctypes.c_int), ("rprod_out", ctypes.c_int)]


def supported_shapes(shapes, skip: Optional[int]) -> bool:
    """shapes: [(out, in)] of lin0..lin_last.  Hidden layers of one width <= 256 (narrower ones run zero-padded to
    256: relu(0) = 0, so the result is unchanged), at most 8 of them, one optional skip layer."""
    n_hidden = len(shapes) - 1
    if not 1 <= n_hidden <= 8:
        return False
    k0, hid = shapes[0][1], shapes[0][0]
    if not 1 <= hid <= _HIDDEN:
        return False
    for l, (o, i) in enumerate(shapes[:-1]):
        if o != hid:
            return False
        want_in = k0 if l == 0 else (hid + k0 if l == skip else hid)
        if i != want_in:
            return False
    n_out, last_in = shapes[-1]
    if last_in != hid or n_out < 1 or (n_out > 4 and n_out % 4):
        return False
    return skip is None or 1 <= skip < n_hidden


def _pad_to_engine_width(ws, bs, skip):
    """Zero-pad hidden layers narrower than 256 to the engine's width (rows of every hidden layer, the matching
    input columns of the next one)."""
    hid = ws[0].shape[0]
    if hid == _HIDDEN:
        return ws, bs
    k0 = ws[0].shape[1]
    out_w, out_b = [], []
    for l, (w, b) in enumerate(zip(ws, bs)):
        last = l == len(ws) - 1
        rows = w.shape[0] if last else _HIDDEN
        if l == 0:
            wp = w.new_zeros(rows, k0)
            wp[:w.shape[0]] = w
        else:
            cols = _HIDDEN + (k0 if l == skip else 0)
            wp = w.new_zeros(rows, cols)
            wp[:w.shape[0], :hid] = w[:, :hid]
            if l == skip:
                wp[:w.shape[0], _HIDDEN:] = w[:, hid:]
        bp = b if last else torch.cat([b, b.new_zeros(_HIDDEN - hid)])
        out_w.append(wp.contiguous())
        out_b.append(bp.contiguous())
    return out_w, out_b


class PackedMlp:
    def __init__(self, weights: List[torch.Tensor], biases: List[torch.Tensor], skip: Optional[int] = None):
        L = _lib.lib()
        shapes = [tuple(w.shape) for w in weights]
        assert supported_shapes(shapes, skip), f"unsupported MLP for the HIP path: {shapes}, skip {skip}"
        dev = weights[0].device
        assert dev.type == "cuda"
        ws = [w.detach().contiguous().float() for w in weights]
        bs = [b.detach().contiguous().float() for b in biases]
        self.true_hidden = shapes[0][0]
        ws, bs = _pad_to_engine_width(ws, bs, skip)
        self.device = dev
        self.k0 = shapes[0][1]
        self.kin_kb = (self.k0 + 7) // 8
        self.n_hidden = len(ws) - 1
        self.n_out = shapes[-1][0]
        self.skip = skip
        per_kb = L.pir_mlp_product_floats(1)
        # ---- layout of the product region (in floats)
        off = 0
        jobs = []          # (W, ld, row0, nrows, col0, ncols, transposed, scale, float offset, n_kblocks)
        plan = PirMlpPlan()
        plan.kin_kb, plan.n_hidden, plan.skip, plan.n_out = self.kin_kb, self.n_hidden, -1 if skip is None else skip, self.n_out
        is2 = 1.0 / math.sqrt(2.0)

        def add(W, row0, nrows, col0, ncols, transposed, scale, nkb):
            nonlocal off
            jobs.append((W, W.shape[1], row0, nrows, col0, ncols, transposed, scale, off, nkb))
            start = off // 4
            off += per_kb * nkb
            return start

        plan.prod_in = add(ws[0], 0, _HIDDEN, 0, self.k0, 0, 1.0, self.kin_kb)
        for l in range(1, self.n_hidden):
            sc = is2 if l == skip else 1.0
            plan.prod_h[l] = add(ws[l], 0, _HIDDEN, 0, _HIDDEN, 0, sc, _ACT_KB)
        if skip is not None:
            plan.prod_skip_x = add(ws[skip], 0, _HIDDEN, _HIDDEN, self.k0, 0, is2, self.kin_kb)
        passes = (self.n_out + 255) // 256
        if self.n_out > 4:
            plan.prod_out = off // 4
            for p in range(passes):
                add(ws[-1], 256 * p, min(256, self.n_out - 256 * p), 0, _HIDDEN, 0, 1.0, _ACT_KB)
        # transposed products of the backward
        plan.has_bwd = 1 if (self.n_out <= 4 or self.n_out % 8 == 0) else 0
        if plan.has_bwd:
            xpasses = (self.kin_kb + _ACT_KB - 1) // _ACT_KB
            for l in range(1, self.n_hidden):
                plan.rprod_h[l] = add(ws[l], 0, _HIDDEN, 0, _HIDDEN, 1, is2 if l == skip else 1.0, _ACT_KB)
            plan.rprod_in = off // 4
            for p in range(xpasses):
                add(ws[0], 0, min(256, self.k0 - 256 * p), 256 * p, _HIDDEN, 1, 1.0, _ACT_KB)
            if skip is not None:
                plan.rprod_skip_x = off // 4
                for p in range(xpasses):
                    add(ws[skip], 0, min(256, self.k0 - 256 * p), _HIDDEN + 256 * p, _HIDDEN, 1, is2, _ACT_KB)
            if self.n_out > 4:
                plan.rprod_out = add(ws[-1], 0, _HIDDEN, 0, self.n_out, 1, 1.0, self.n_out // 8)
        self.mats_floats = off
        # ---- vectors
        voff = 0
        plan.bias_off = voff
        voff += 256 * self.n_hidden
        if self.n_out <= 4:
            plan.head_off = voff
            voff += 256 * self.n_out
        plan.out_bias_off = voff
        voff += 4 if self.n_out <= 4 else 256 * passes
        self.plan = plan
        self.buf = torch.zeros(self.mats_floats + voff, dtype=torch.float32, device=dev)
        vec = self.buf[self.mats_floats:]
        for l in range(self.n_hidden):
            vec[plan.bias_off + 256 * l: plan.bias_off + 256 * (l + 1)] = bs[l]
        if self.n_out <= 4:
            vec[plan.head_off: plan.head_off + 256 * self.n_out] = ws[-1].reshape(-1)
        vec[plan.out_bias_off: plan.out_bias_off + self.n_out] = bs[-1]
        with torch.cuda.device(dev):
            for (W, ld, row0, nrows, col0, ncols, tr, sc, foff, nkb) in jobs:
                dst = ctypes.c_void_p(self.buf.data_ptr() + 4 * foff)
                _lib.check(L.pir_mlp_pack_product(_lib.ptr(W), ld, row0, nrows, col0, ncols, tr, sc, dst, nkb,
                                                  _lib.stream_ptr()), "pir_mlp_pack_product")
        self._keep = ws   # the pack kernels read them asynchronously

    def padded_shapes(self, shapes):
        """Shapes of the gradient buffers the kernels write, for the parameter shapes `shapes` = the n_hidden + 1
        weights followed by the n_hidden + 1 biases."""
        nl, hid = self.n_hidden + 1, self.true_hidden
        if hid == _HIDDEN:
            return [tuple(s) for s in shapes]
        out = []
        for idx, s in enumerate(shapes):
            s = tuple(s)
            l = idx if idx < nl else idx - nl
            last = l == nl - 1
            if idx < nl:
                out.append((s[0] if last else _HIDDEN, s[1] if l == 0 else _HIDDEN + (s[1] - hid)))
            else:
                out.append(s if last else (_HIDDEN,))
        return out

    def unpad_grads(self, grads, shapes):
        hid = self.true_hidden
        if hid == _HIDDEN:
            return grads
        out = []
        for g, s in zip(grads, shapes):
            s = tuple(s)
            if g.shape == s:
                out.append(g)
            elif g.dim() == 1:
                out.append(g[:s[0]].contiguous())
            elif g.shape[1] == s[1]:
                out.append(g[:s[0]].contiguous())
            else:   # [256, 256 (+ k0)] -> [hid, hid (+ k0)]
                out.append(torch.cat([g[:s[0], :hid], g[:s[0], _HIDDEN:]], dim=1).contiguous())
        return out

    def pad_input(self, x: torch.Tensor) -> torch.Tensor:
        """[P, k0] -> contiguous [P, 8 kin_kb] with zero columns appended."""
        x = x.detach().reshape(-1, self.k0).float()
        pad = 8 * self.kin_kb - self.k0
        if pad:
            x = torch.cat([x, x.new_zeros(x.shape[0], pad)], dim=-1)
        return x.contiguous()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """Pre-transform output [P, n_out] of the last linear layer for input rows x [P, k0]."""
        return self.forward_padded(self.pad_input(x))

    def forward_padded(self, xp: torch.Tensor) -> torch.Tensor:
        P = xp.shape[0]
        out = torch.empty(P, self.n_out, dtype=torch.float32, device=xp.device)
        with torch.cuda.device(xp.device):
            _lib.check(_lib.lib().pir_mlp_forward(_lib.ptr(self.buf), self.mats_floats, ctypes.byref(self.plan),
                                                  _lib.ptr(xp), P, _lib.ptr(out), _lib.stream_ptr()), "pir_mlp_forward")
        return out


def module_layers(mod):
    return [getattr(mod, "lin" + str(l)) for l in range(mod.num_layers - 1)]


def module_supported(mod) -> bool:
    p = next(mod.parameters())
    if not p.is_cuda or p.dtype != torch.float32 or len(mod.skip_in) > 1:
        return False
    lins = module_layers(mod)
    shapes = [tuple(l.weight_v.shape) if l.weight_norm else tuple(l.weight.shape) for l in lins]
    return supported_shapes(shapes, mod.skip_in[0] if mod.skip_in else None)


def packed_for(mod) -> PackedMlp:
    """Packed weights of a RenderingNetwork / DINONetwork, rebuilt whenever a parameter changed."""
    key = tuple((p.data_ptr(), p._version) for p in mod.parameters())
    if getattr(mod, "_packed_mlp", None) is None or mod._packed_mlp_key != key:
        with torch.no_grad():
            lins = module_layers(mod)
            mod._packed_mlp = PackedMlp([l.effective_weight() for l in lins], [l.bias for l in lins],
                                        mod.skip_in[0] if mod.skip_in else None)
        mod._packed_mlp_key = key
    return mod._packed_mlp


def mlp_apply_eager(mod, inp):
    """The eager PyTorch expression (rocBLAS GEMMs): the definition of the result."""
    import numpy as np
    import torch.nn.functional as F
    x = inp
    for l in range(mod.num_layers - 1):
        if l in mod.skip_in:
            x = torch.cat([x, inp], dim=-1) / np.sqrt(2)
        x = getattr(mod, "lin" + str(l))(x)
        if l < mod.num_layers - 2:
            x = F.relu(x)
    return x


def _eager_from_weights(inp, wb, skip):
    """mlp_apply_eager on explicit effective weights / biases (a differentiable function of all of them)."""
    import numpy as np
    import torch.nn.functional as F
    n = len(wb) // 2
    x = inp
    for l in range(n):
        if skip is not None and l == skip:
            x = torch.cat([x, inp], dim=-1) / np.sqrt(2)
        x = F.linear(x, wb[l], wb[n + l])
        if l < n - 1:
            x = F.relu(x)
    return x


_bwd_scratch_cache = {}
BWD_CHUNK_POINTS = 131072
_force_eager = 0


class eager_grad:
    """Context manager: differentiable evaluations inside use the eager PyTorch expression (A/B tests; callers that
    need derivatives the fused path does not provide, i.e. gradients of parameter gradients)."""

    def __enter__(self):
        global _force_eager
        _force_eager += 1

    def __exit__(self, *exc):
        global _force_eager
        _force_eager -= 1
        return False


def _backward_scratch(device, n_hidden, points):
    """Backward scratch of the material MLPs: one buffer per (device, stream), sized to the largest call seen there
    (points rounded up to a power of two, at most BWD_CHUNK_POINTS).  The SDF backward's buffer of the same stream is
    used instead when it is large enough (the two never run concurrently on one stream)."""
    skey = _lib.stream_key(device)
    want = min(BWD_CHUNK_POINTS, max(4096, 1 << max(0, int(points) - 1).bit_length()))
    need = _lib.lib().pir_mlp_backward_scratch_bytes(n_hidden, want)
    from . import sdf_hip
    shared = sdf_hip._bwd_scratch_cache.get(skey)      # same device AND same stream only
    if shared is not None and shared.numel() >= need:
        return shared
    buf = _bwd_scratch_cache.get(skey)
    if buf is None or buf.numel() < need:
        _bwd_scratch_cache.pop(skey, None)
        buf = None
        buf = torch.empty(need, dtype=torch.uint8, device=device)
        _bwd_scratch_cache[skey] = buf
    return buf


def free_scratch():
    """Drop the cached backward scratch buffers of this module (see sdf_hip.free_scratch)."""
    _bwd_scratch_cache.clear()


def _zero_grads(pshapes, dev):
    """One zero-filled allocation viewed as the per-layer gradient tensors (one fill launch instead of one each)."""
    sizes = [int(torch.Size(tuple(s)).numel()) for s in pshapes]
    flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
    return [p.view(tuple(s)) for p, s in zip(torch.split(flat, sizes), pshapes)]


def _run_backward(packed, xp, g_out, need_gx, shapes):
    P = xp.shape[0]
    dev = xp.device
    nl = packed.n_hidden + 1
    pshapes = packed.padded_shapes(shapes)
    zs = _zero_grads(pshapes, dev)
    gws, gbs = zs[:nl], zs[nl:]
    gx = torch.empty(P, xp.shape[1], dtype=torch.float32, device=dev) if need_gx else None
    g = g_out.detach().reshape(P, packed.n_out).contiguous().float()
    scratch = _backward_scratch(dev, packed.n_hidden, P)
    wp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in gws])
    bp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in gbs])
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().pir_mlp_backward(
            _lib.ptr(packed.buf), packed.mats_floats, ctypes.byref(packed.plan), packed.k0, _lib.ptr(xp), P,
            _lib.ptr(g), _lib.ptr(gx), wp, bp, _lib.ptr(scratch), scratch.numel(), _lib.stream_ptr()),
            "pir_mlp_backward")
    gws, gbs = packed.unpad_grads(gws, shapes[:nl]), packed.unpad_grads(gbs, shapes[nl:])
    return gx, gws, gbs


class _MlpFn(torch.autograd.Function):
    """out = fused MLP(inp) on the packed effective weights; backward = pir_mlp_backward (recompute + sweeps +
    split-K weight-gradient GEMM).  PyTorch chains dW_l -> (weight_g, weight_v) through torch._weight_norm.
    Under create_graph=True the backward is itself recorded (`_MlpBwdFn`), so the input gradient can be
    differentiated once more (the smoothness loss, render_surface.py:40-48)."""

    @staticmethod
    def forward(ctx, mod, inp, *wb):
        packed = packed_for(mod)
        xp = packed.pad_input(inp.reshape(-1, inp.shape[-1]))
        out = packed.forward_padded(xp)
        ctx.packed, ctx.xp, ctx.in_shape = packed, xp, inp.shape
        ctx.need_gx = inp.requires_grad
        ctx.skip = mod.skip_in[0] if mod.skip_in else None
        ctx.save_for_backward(inp, *wb)
        return out.reshape(*inp.shape[:-1], out.shape[-1])

    @staticmethod
    def backward(ctx, g_out):
        inp, *wb = ctx.saved_tensors
        packed, xp = ctx.packed, ctx.xp
        if torch.is_grad_enabled():          # create_graph=True: the gradients must stay differentiable
            if ctx.need_gx and packed.n_out <= 4:     # fused second order (tangent kernel) for d/d input
                res = _MlpBwdFn.apply(packed, xp, ctx.in_shape, g_out, *wb)
                return (None, res[0], *res[1:])
            # no fused second-order path for this shape (wide outputs such as the 384-wide DINO field, or second
            # derivatives w.r.t. the weights only): differentiate the eager expression so that nothing is silently
            # cut out of the graph
            with torch.enable_grad():
                out = _eager_from_weights(inp, wb, ctx.skip)
                wanted = [t for t in [inp] + list(wb) if t.requires_grad]
                got = torch.autograd.grad(out, wanted, g_out.reshape(out.shape), create_graph=True, allow_unused=True)
            it = iter(got)
            grads = [next(it) if t.requires_grad else None for t in [inp] + list(wb)]
            return (None, *grads)
        gx, gws, gbs = _run_backward(packed, xp, g_out, ctx.need_gx, [tuple(t.shape) for t in wb])
        g_in = gx[:, :packed.k0].reshape(ctx.in_shape) if ctx.need_gx else None
        return (None, g_in, *gws, *gbs)


class _MlpBwdFn(torch.autograd.Function):
    """(g_out, W, b) -> (g_in, gW, gb) of the fused MLP as a differentiable node.  Its backward handles an upstream
    gradient on g_in (pir_mlp_backward_tangent); upstream gradients on gW / gb are not supported (the reference
    never differentiates parameter gradients)."""

    @staticmethod
    def forward(ctx, packed, xp, in_shape, g_out, *wb):
        gx, gws, gbs = _run_backward(packed, xp, g_out, True, [tuple(t.shape) for t in wb])
        ctx.packed, ctx.xp, ctx.in_shape = packed, xp, in_shape
        ctx.g_out = g_out.detach().reshape(xp.shape[0], packed.n_out).contiguous().float()
        ctx.g_out_shape = g_out.shape
        ctx.shapes = [tuple(t.shape) for t in wb]
        ctx.set_materialize_grads(False)
        return (gx[:, :packed.k0].reshape(in_shape), *gws, *gbs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gbar_in, *gbar_wb):
        if any(g is not None for g in gbar_wb):
            raise NotImplementedError("pir_amd: gradients of the material networks' parameter gradients are not supported")
        nl = ctx.packed.n_hidden + 1
        if gbar_in is None:
            return (None,) * (4 + 2 * nl)
        packed, xp = ctx.packed, ctx.xp
        P, dev = xp.shape[0], xp.device
        gbar = packed.pad_input(gbar_in.reshape(P, packed.k0))
        ghat = torch.empty(P, packed.n_out, dtype=torch.float32, device=dev)
        gws = _zero_grads(packed.padded_shapes(ctx.shapes)[:nl], dev)
        scratch = _backward_scratch(dev, packed.n_hidden, P)
        wp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in gws])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().pir_mlp_backward_tangent(
                _lib.ptr(packed.buf), packed.mats_floats, ctypes.byref(packed.plan), packed.k0, _lib.ptr(xp), P,
                _lib.ptr(ctx.g_out), _lib.ptr(gbar), _lib.ptr(ghat), wp, _lib.ptr(scratch), scratch.numel(),
                _lib.stream_ptr()), "pir_mlp_backward_tangent")
        gws = packed.unpad_grads(gws, ctx.shapes[:nl])
        return (None, None, None, ghat.reshape(ctx.g_out_shape), *gws, *([None] * nl))


def mlp_apply_fused(mod, inp):
    lins = module_layers(mod)
    # shared_weight(): ONE weight-norm node per layer and optimisation step, shared by every evaluation of the network
    # (direct shading, both indirect shades, the edge sides): no weight-norm launches per call, one backward per step
    wb = [l.shared_weight() for l in lins] + [l.bias for l in lins]
    return _MlpFn.apply(mod, inp, *wb)


def mlp_apply(mod, inp):
    """Differentiable evaluation: the fused kernels when the network has a fused backward (<= 4 outputs) and the
    caller has not asked for the twice-differentiable eager expression."""
    if _force_eager or packed_for(mod).plan.has_bwd == 0:
        return mlp_apply_eager(mod, inp)
    return mlp_apply_fused(mod, inp)
