"""torch.autograd.Function wrappers that put the HIP SDF kernels into the autograd graph.

forward  : pir_sdf_forward / pir_sdf_get_all on the packed weights
backward : pir_sdf_get_all_backward -> gradients of the EFFECTIVE weights W_l = g v/|v| and biases; PyTorch then
           chains dW_l -> (weight_g, weight_v) through torch._weight_norm's own backward, and the x gradient on
           to whatever produced the points (light position, reparameterised hit points).
The functions are once-differentiable (the reference never differentiates the training loss twice).
"""
import torch

from . import sdf_hip

SKIP_ZERO_MIN_POINTS = 1 << 16   # below this the mask + host read-back costs more than it can save
SKIP_ZERO_FRACTION = 0.9         # compact when fewer than this fraction of the points carry gradient
stats = {}


class _SdfGetAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, want_feature, want_normal, x, theta):
        packed = net.packed()
        if want_normal:
            sdf, feat, nrm = packed.get_all(x, want_feature=want_feature)
        else:
            out = packed.forward(x, want_feature=want_feature)
            sdf, feat = out if want_feature else (out, None)
            nrm = None
        ctx.packed = packed
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x)
        ctx.need_gx = x.requires_grad
        ctx.want = (want_feature, want_normal)
        outs = [sdf, feat if want_feature else x.new_zeros(0), nrm if want_normal else x.new_zeros(0)]
        non_diff = [o for o, keep in zip(outs[1:], ctx.want) if not keep]
        if non_diff:
            ctx.mark_non_differentiable(*non_diff)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_sdf, g_feat, g_nrm):
        (x,) = ctx.saved_tensors
        want_feature, want_normal = ctx.want
        g_feat = g_feat if want_feature else None
        g_nrm = g_nrm if want_normal else None
        # Points whose upstream gradients are all exactly zero contribute exactly nothing (Zb = 0 and the tangent
        # is 0, so both weight-gradient products and dL/dx vanish): drop them before the sweeps.  In the light
        # march these are the samples whose sigmoid saturated in fp32 (most of free space for a trained inv_s).
        M = x.shape[0]
        sel = None
        if g_feat is None and M >= SKIP_ZERO_MIN_POINTS and (g_sdf is not None or g_nrm is not None):
            nz = torch.zeros(M, dtype=torch.bool, device=x.device)
            if g_sdf is not None:
                nz |= g_sdf.reshape(M) != 0
            if g_nrm is not None:
                nz |= (g_nrm.reshape(M, 3) != 0).any(dim=-1)
            n_nz = int(nz.sum().item())
            stats["last_nonzero_fraction"] = n_nz / max(1, M)
            if n_nz < SKIP_ZERO_FRACTION * M:
                sel = nz.nonzero(as_tuple=True)[0]
        if sel is not None:
            pick = lambda g, w: None if g is None else g.reshape(M, *w).index_select(0, sel)  # noqa: E731
            gws, gbs, gx_sel = sdf_hip.get_all_backward(ctx.packed, x.index_select(0, sel), pick(g_sdf, ()),
                                                        None, pick(g_nrm, (3,)), ctx.need_gx)
            gx = None
            if ctx.need_gx:
                gx = torch.zeros_like(x).index_copy_(0, sel, gx_sel)
        else:
            gws, gbs, gx = sdf_hip.get_all_backward(ctx.packed, x, g_sdf, g_feat, g_nrm, ctx.need_gx)
        # gws / gbs are consecutive views of one flat allocation laid out like theta: hand it back as one gradient
        return (None, None, None, gx, sdf_hip.flat_of(gws, gbs))


def _theta(net):
    """The network's parameters as the kernels' single autograd input.  Weight-normed layers only (the reference's
    configuration, src/network.py:75); theta() caches the node until a parameter changes."""
    return net.theta()


def sdf_get_all(net, x, want_feature=True):
    sdf, feat, nrm = _SdfGetAll.apply(net, want_feature, True, x, _theta(net))
    return sdf, (feat if want_feature else None), nrm


def sdf_forward(net, x):
    sdf, feat, _ = _SdfGetAll.apply(net, True, False, x, _theta(net))
    return sdf, feat
