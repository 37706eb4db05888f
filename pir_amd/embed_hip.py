"""Host side of the fused frequency-encoding kernels (pir_embed_* in include/pir_hip.h; reference:
src/models/embedder.py:20-41).  One launch per encoding and per backward instead of 3 L + 1 eager launches each way;
twice differentiable, because the material smoothness loss differentiates the networks w.r.t. the surface point with
create_graph=True (render_surface.py:40-48) before loss.backward() runs."""
import torch
from torch.autograd.function import once_differentiable

from . import _lib


def _call(name, P, D, L, *tensors):
    with torch.cuda.device(tensors[0].device):
        _lib.check(getattr(_lib.lib(), name)(P, D, L, *[_lib.ptr(t) for t in tensors], _lib.stream_ptr()), name)


class _EmbedBackward(torch.autograd.Function):
    """(x, g_out) -> g_x, differentiable once more."""

    @staticmethod
    def forward(ctx, x, g_out, L):
        P, D = x.shape
        g_out = g_out.contiguous()
        g_x = torch.empty_like(x)
        _call("pir_embed_backward", P, D, L, x, g_out, g_x)
        ctx.save_for_backward(x, g_out)
        ctx.L = L
        return g_x

    @staticmethod
    @once_differentiable
    def backward(ctx, u):
        x, g_out = ctx.saved_tensors
        P, D = x.shape
        u = u.contiguous()
        gx = gg = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty_like(x)
            _call("pir_embed_second", P, D, ctx.L, x, g_out, u, gx)
        if ctx.needs_input_grad[1]:
            gg = torch.empty_like(g_out)
            _call("pir_embed_tangent", P, D, ctx.L, x, u, gg)
        return gx, gg, None


class _Embed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, L):
        P, D = x.shape
        out = torch.empty(P, D * (1 + 2 * L), dtype=torch.float32, device=x.device)
        _call("pir_embed_forward", P, D, L, x, out)
        ctx.save_for_backward(x)
        ctx.L = L
        return out

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return None, None
        return _EmbedBackward.apply(x, g, ctx.L), None


def supported(x, n_freqs, input_dims):
    return (x.is_cuda and x.dtype == torch.float32 and x.shape[-1] == input_dims and 1 <= input_dims <= 16
            and 0 <= n_freqs <= 16)


def embed(x, n_freqs):
    """[..., D] -> [..., D (1 + 2 n_freqs)] on the HIP kernels."""
    lead = x.shape[:-1]
    flat = x.reshape(-1, x.shape[-1])
    if not flat.is_contiguous():
        flat = flat.contiguous()
    width = x.shape[-1] * (1 + 2 * int(n_freqs))
    return _Embed.apply(flat, int(n_freqs)).reshape(*lead, width)
