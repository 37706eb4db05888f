"""torch.autograd.Function wrappers that put the HIP SDF kernels into the autograd graph.

forward  : pir_sdf_forward / pir_sdf_get_all on the packed weights
backward : pir_sdf_get_all_backward -> gradients of the EFFECTIVE weights W_l = g v/|v| and biases; PyTorch then
           chains dW_l -> (weight_g, weight_v) through torch._weight_norm's own backward, and the x gradient on
           to whatever produced the points (light position, reparameterised hit points).
The functions are once-differentiable (the reference never differentiates the training loss twice).
"""
import torch

from . import sdf_hip


class _SdfGetAll(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, want_feature, want_normal, x, *wb):
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
        gws, gbs, gx = sdf_hip.get_all_backward(ctx.packed, x, g_sdf, g_feat if want_feature else None,
                                                g_nrm if want_normal else None, ctx.need_gx)
        return (None, None, None, gx, *gws, *gbs)


def _wb(net):
    ws, bs = net.effective_weights()
    return list(ws) + list(bs)


def sdf_get_all(net, x, want_feature=True):
    sdf, feat, nrm = _SdfGetAll.apply(net, want_feature, True, x, *_wb(net))
    return sdf, (feat if want_feature else None), nrm


def sdf_forward(net, x):
    sdf, feat, _ = _SdfGetAll.apply(net, True, False, x, *_wb(net))
    return sdf, feat
