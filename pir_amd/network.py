"""Drop-in counterparts of the reference's src/network.py modules.

`SDFNetwork` keeps the reference constructor, methods (`forward/sdf/sdf_hidden_appearance/gradient/get_all`)
and state-dict keys (`lin{l}.weight_g`, `lin{l}.weight_v`, `lin{l}.bias`, src/network.py:8-142).  On a CUDA
(ROCm) device with the configs/model_config.json architecture every evaluation runs in the HIP kernels of
libpir_hip.so (pir_amd/csrc); the eager PyTorch expression below is the definition used on CPU (host-logic
tests) and for non-standard architectures.  The material / light / DINO-field modules are the PyTorch
(rocBLAS) side of the path.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .models.embedder import get_embedder
from . import sdf_autograd


class _WeightNormFn(torch.autograd.Function):
    """W = g v / |v| (rows) as ONE graph node that can be shared by every evaluation of a network within an
    optimisation step and traversed by any number of backward calls: it keeps v, g and the row norms as plain
    attributes instead of autograd-saved tensors (which a first backward would free).  Same kernels as
    torch._weight_norm (aten::_weight_norm_interface and its backward)."""

    @staticmethod
    def forward(ctx, v, g):
        w, norms = torch._weight_norm_interface(v, g, 0)
        ctx.v, ctx.g, ctx.norms = v, g, norms
        ctx.versions = (v._version, g._version)
        return w

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gw):
        if (ctx.v._version, ctx.g._version) != ctx.versions:
            raise RuntimeError("pir_amd: a weight-normed parameter was modified in place between the forward pass and "
                               "the backward pass that differentiates it")
        gv, gg = torch.ops.aten._weight_norm_interface_backward(gw.contiguous(), ctx.v, ctx.g, ctx.norms, 0)
        return gv, gg


class _SdfThetaFn(torch.autograd.Function):
    """All effective weights and biases of an SDFNetwork as ONE flat tensor theta = [W_0 .. W_8 | b_0 .. b_8]: every
    kernel call of a step takes theta as its single parameter input and returns a single flat gradient, so autograd
    adds one tensor per call instead of 27 and runs the weight-norm backward once per step instead of once per call.
    Like _WeightNormFn it holds its inputs as attributes, so several backward calls may traverse it."""

    @staticmethod
    def forward(ctx, *vgb):
        n = len(vgb) // 3
        vs, gs, bs = vgb[:n], vgb[n:2 * n], vgb[2 * n:]
        ws, norms = [], []
        for v, g in zip(vs, gs):
            w, nr = torch._weight_norm_interface(v, g, 0)
            ws.append(w.reshape(-1))
            norms.append(nr)
        ctx.vs, ctx.gs, ctx.norms = vs, gs, norms
        ctx.versions = tuple(t._version for t in vs + gs)
        ctx.sizes = [w.numel() for w in ws] + [b.numel() for b in bs]
        return torch.cat(ws + [b.reshape(-1) for b in bs])

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_theta):
        if tuple(t._version for t in ctx.vs + ctx.gs) != ctx.versions:
            raise RuntimeError("pir_amd: an SDF parameter was modified in place between the forward pass and the "
                               "backward pass that differentiates it")
        n = len(ctx.vs)
        parts = torch.split(g_theta, ctx.sizes)
        gvs, ggs = [], []
        for l in range(n):
            gv, gg = torch.ops.aten._weight_norm_interface_backward(parts[l].view(ctx.vs[l].shape), ctx.vs[l], ctx.gs[l],
                                                                    ctx.norms[l], 0)
            gvs.append(gv)
            ggs.append(gg)
        return (*gvs, *ggs, *[parts[n + l].view(-1) for l in range(n)])


class WNLinear(nn.Module):
    """Linear layer with old-style weight normalisation along dim 0 (parameters weight_g [out,1],
    weight_v [out,in], bias [out]) -- the parameter set nn.utils.weight_norm(nn.Linear) exposes."""

    def __init__(self, d_in, d_out, weight_norm=True):
        super().__init__()
        lin = nn.Linear(d_in, d_out)  # PyTorch's default init, then re-parameterise
        self.weight_norm = weight_norm
        self.bias = nn.Parameter(lin.bias.detach().clone())
        if weight_norm:
            w = lin.weight.detach()
            self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True))
            self.weight_v = nn.Parameter(w.clone())
        else:
            self.weight = nn.Parameter(lin.weight.detach().clone())

    def set_weight(self, w):
        """Re-initialise from a plain weight matrix (g = row norms, v = w)."""
        with torch.no_grad():
            if self.weight_norm:
                self.weight_v.copy_(w)
                self.weight_g.copy_(w.norm(dim=1, keepdim=True))
            else:
                self.weight.copy_(w)

    def effective_weight(self):
        if not self.weight_norm:
            return self.weight
        return torch._weight_norm(self.weight_v, self.weight_g, 0)

    def shared_weight(self):
        """effective_weight() as a graph node shared by all calls until a parameter changes (GPU kernel paths: one
        weight-norm launch per optimisation step instead of one per network evaluation, and one weight-norm backward)."""
        if not self.weight_norm:
            return self.weight
        key = (self.weight_v.data_ptr(), self.weight_v._version, self.weight_g.data_ptr(), self.weight_g._version,
               self.weight_v.requires_grad, self.weight_g.requires_grad, torch.is_grad_enabled())
        if getattr(self, "_shared_key", None) != key:
            self._shared_w = _WeightNormFn.apply(self.weight_v, self.weight_g)
            self._shared_key = key
        return self._shared_w

    def forward(self, x):
        return F.linear(x, self.effective_weight(), self.bias)


def invalidate_packed(module):
    """invalidate_packed for any pir_amd network (SDFNetwork, RenderingNetwork, BlendNetwork, DINONetwork) or a
    container of them: needed only after writes through `.data`, which the version-keyed caches cannot see."""
    for m in module.modules():
        if isinstance(m, SDFNetwork):
            m.invalidate_packed()
        if getattr(m, "_packed_mlp", None) is not None:
            m._packed_mlp = None
            m._packed_mlp_key = None


def _relu_mlp(mod, inp, keep_graph=True):
    """lin0..lin_last of a RenderingNetwork / BlendNetwork / DINONetwork on the concatenated input rows `inp`
    ([..., k0]; skip layers read [x | inp] / sqrt 2, ReLU after all but the last layer; src/network.py:232-243).
    On a GPU, for the 256-wide architectures of configs/model_config.json, all layers run in one fused HIP
    kernel (pir_amd/csrc/mlp.hip); the eager expression below defines the result everywhere else.
    keep_graph=False: the caller detaches the result (DINONetwork with is_training=False)."""
    from . import mlp_hip
    if inp.is_cuda and mlp_hip.module_supported(mod):
        needs_grad = keep_graph and torch.is_grad_enabled() and (
            inp.requires_grad or any(p.requires_grad for p in mod.parameters()))
        if not needs_grad:
            out = mlp_hip.packed_for(mod).forward(inp.reshape(-1, inp.shape[-1]))
            return out.reshape(*inp.shape[:-1], out.shape[-1])
        return mlp_hip.mlp_apply(mod, inp)
    x = inp
    for l in range(mod.num_layers - 1):
        if l in mod.skip_in:
            x = torch.cat([x, inp], dim=-1) / np.sqrt(2)
        x = getattr(mod, "lin" + str(l))(x)
        if l < mod.num_layers - 2:
            x = F.relu(x)
    return x


class SDFNetwork(nn.Module):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        dims = [d_in] + [d_hidden] * n_layers + [d_out]
        self.embed_fn_fine = None
        if multires > 0:
            self.embed_fn_fine, dims[0] = get_embedder(multires, input_dims=d_in)
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.multires = multires
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = WNLinear(dims[l], out_dim, weight_norm)
            if geometric_init:  # IDR geometric initialisation (src/network.py:46-72): a sphere of radius `bias`
                w = torch.empty(out_dim, dims[l])
                if l == self.num_layers - 2:
                    mean = math.sqrt(math.pi) / math.sqrt(dims[l])
                    nn.init.normal_(w, mean=-mean if inside_outside else mean, std=1e-4)
                    nn.init.constant_(lin.bias, bias if inside_outside else -bias)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(w, 0.0, math.sqrt(2) / math.sqrt(out_dim))
                    if multires > 0 and l == 0:
                        w[:, 3:] = 0.0
                    elif multires > 0 and l in self.skip_in:
                        w[:, -(dims[0] - 3):] = 0.0
                lin.set_weight(w)
            setattr(self, "lin" + str(l), lin)
        self.activation = nn.Softplus(beta=100)
        self._packed = None
        self._packed_key = None

    # ------------------------------------------------------------------ parameters for the kernels
    def layers(self):
        return [getattr(self, "lin" + str(l)) for l in range(self.num_layers - 1)]

    def effective_weights(self):
        return [lin.effective_weight() for lin in self.layers()], [lin.bias for lin in self.layers()]

    def theta(self):
        """[W_0 .. W_8 | b_0 .. b_8] flat, with its graph to the parameters, shared by all kernel calls until a parameter
        changes (_SdfThetaFn)."""
        lins = self.layers()
        ps = [l.weight_v for l in lins] + [l.weight_g for l in lins] + [l.bias for l in lins]
        key = tuple((p.data_ptr(), p._version, p.requires_grad) for p in ps) + (torch.is_grad_enabled(),)
        if getattr(self, "_theta_key", None) != key:
            self._theta = _SdfThetaFn.apply(*ps)
            self._theta_key = key
        return self._theta

    def hip_supported(self):
        from .sdf_hip import supported_architecture
        p = next(self.parameters())
        return (p.is_cuda and self.scale == 1 and self.skip_in == (4,) and self.multires == 6
                and supported_architecture([tuple(l.effective_weight().shape) if not l.weight_norm else
                                            tuple(l.weight_v.shape) for l in self.layers()]))

    def packed(self):
        """Packed weights for the HIP kernels, rebuilt whenever a parameter changed (optimizer step, load)."""
        from .sdf_hip import PackedSdf
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is None or key != self._packed_key:
            with torch.no_grad():
                ws, bs = self.effective_weights()
                self._packed = PackedSdf(ws, bs)
            self._packed_key = key
        return self._packed

    def invalidate_packed(self):
        """Drop the packed weights.  The cache key is (data_ptr, version) of every parameter, which in-place writes
        through autograd-visible tensors bump (optimizers, load_state_dict); writes through `.data` do NOT, so code
        that edits `p.data` directly (EMA swaps, manual re-initialisation) must call this afterwards."""
        self._packed = None
        self._packed_key = None

    # ------------------------------------------------------------------ reference API
    def _forward_torch(self, inputs):
        inputs = inputs * self.scale
        e = self.embed_fn_fine(inputs) if self.embed_fn_fine is not None else inputs
        x = e
        for l, lin in enumerate(self.layers()):
            if l in self.skip_in:
                x = torch.cat([x, e], -1) / np.sqrt(2)
            x = lin(x)
            if l < self.num_layers - 2:
                x = self.activation(x)
        return torch.cat([x[..., :1] / self.scale, x[..., 1:]], dim=-1)

    def forward(self, inputs):
        if self.hip_supported():
            shp = inputs.shape[:-1]
            needs_grad = torch.is_grad_enabled() and (inputs.requires_grad or any(p.requires_grad for p in self.parameters()))
            if not needs_grad:
                sdf, feat = self.packed().forward(inputs.reshape(-1, 3), want_feature=True)
                return torch.cat([sdf[:, None], feat], dim=-1).reshape(*shp, -1)
            sdf, feat = sdf_autograd.sdf_forward(self, inputs.reshape(-1, 3))
            return torch.cat([sdf[:, None], feat], dim=-1).reshape(*shp, -1)
        return self._forward_torch(inputs)

    def sdf(self, x):
        if self.hip_supported() and not (torch.is_grad_enabled() and (x.requires_grad or
                                         any(p.requires_grad for p in self.parameters()))):
            return self.packed().forward(x.reshape(-1, 3)).reshape(*x.shape[:-1], 1)
        return self.forward(x)[..., :1]

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        return self.get_all(x, is_training=True)[2]

    def get_all(self, x, is_training=True):
        if self.hip_supported():
            x.requires_grad_(True)  # the reference marks its argument in place (src/network.py:127)
            if not is_training:
                sdf, feat, nrm = self.packed().get_all(x.reshape(-1, 3), want_feature=True)
                return sdf[:, None], feat, nrm
            sdf, feat, nrm = sdf_autograd.sdf_get_all(self, x.reshape(-1, 3), want_feature=True)
            return sdf[:, None], feat, nrm
        with torch.enable_grad():
            x.requires_grad_(True)
            out = self._forward_torch(x)
            y, feature = out[..., :1], out[..., 1:]
            (g,) = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=is_training, retain_graph=is_training)
        if not is_training:
            return y.detach(), feature.detach(), g.detach()
        return y, feature, g

    def get_sdf_normal(self, x, is_training=True):
        """sdf [M], normal [M,3] without the 256 features (the light-visibility march needs no features)."""
        if self.hip_supported():
            if not is_training:
                sdf, _, nrm = self.packed().get_all(x.reshape(-1, 3), want_feature=False)
                return sdf, nrm
            sdf, _, nrm = sdf_autograd.sdf_get_all(self, x.reshape(-1, 3), want_feature=False)
            return sdf, nrm
        y, _, g = self.get_all(x, is_training)
        return y[..., 0], g


class RenderingNetwork(nn.Module):
    """Material MLP (reference: src/network.py:146-267).  mode 'idr': [pts, view, normal, geo_f];
    mode 'no_view_dir': [pts, normal, geo_f, semantic_f]."""

    def __init__(self, mode, d_in, d_feature, d_out, d_hidden, n_layers, weight_norm=True, multires=0,
                 multires_view=0, squeeze_out=True, squeeze_out_scale=1.0, output_bias=0.0, output_scale=1.0,
                 skip_in=()):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        dims = [d_in + d_feature] + [d_hidden] * n_layers + [d_out]
        self.embed_fn = self.embedview_fn = None
        if multires > 0:
            self.embed_fn, ch = get_embedder(multires)
            dims[0] += ch - 3
        if multires_view > 0:
            self.embedview_fn, ch = get_embedder(multires_view)
            dims[0] += ch - 3
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                dims[l] += dims[0]
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            setattr(self, "lin" + str(l), WNLinear(dims[l], out_dim, weight_norm))
        self.output_bias, self.output_scale, self.squeeze_out_scale = output_bias, output_scale, squeeze_out_scale

    def forward(self, points, normals, view_dirs, geo_f, semantic_f=None):
        if self.embed_fn is not None and points is not None:
            points = self.embed_fn(points)
        if self.embedview_fn is not None:
            view_dirs = self.embedview_fn(view_dirs)
        if self.mode == "idr":
            inp = torch.cat([points, view_dirs, normals, geo_f], dim=-1)
        elif self.mode == "no_view_dir":
            inp = torch.cat([points, normals, geo_f, semantic_f], dim=-1)
        else:
            raise ValueError(self.mode)
        x = _relu_mlp(self, inp)
        x = self.output_scale * (x + self.output_bias)
        if self.squeeze_out:
            x = self.squeeze_out_scale * torch.sigmoid(x)
        return x

    @torch.enable_grad()
    def get_all(self, points, normals, view_dirs, geo_f, semantic_f=None, is_training=True):
        semantic_f.requires_grad_(True)
        y = self.forward(points, normals, view_dirs, geo_f, semantic_f)
        (g,) = torch.autograd.grad(y, semantic_f, torch.ones_like(y), create_graph=is_training, retain_graph=is_training)
        return (y, g) if is_training else (y.detach(), g.detach())


class PointLightNetwork(nn.Module):
    """Learnable light intensity + camera-frame offset (reference: src/network.py:270-292)."""

    def __init__(self, enable_offset=False):
        super().__init__()
        self.light = nn.Parameter(torch.tensor(5.0))
        if enable_offset:
            self.light_offset = nn.Embedding(1, 3)
            self.light_offset.weight.data = torch.zeros(1, 3)

    def forward(self):
        return self.light

    @property
    def offset(self):
        w = self.light_offset.weight
        return self.light_offset(torch.zeros(1, dtype=torch.long, device=w.device))

    def set_light(self, light):
        self.light.data.fill_(light)

    def get_light(self):
        return self.light.data.detach()

    def get_offset(self):
        return self.light_offset.weight.data.detach()


class BlendNetwork(nn.Module):
    """Blend weight of the one-bounce indirect term from (distance, roughness, PE(n.wo))
    (reference: src/network.py:295-367).  squeeze_out=False gives relu(x) * 2 pi."""

    def __init__(self, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=8, squeeze_out=True,
                 squeeze_out_scale=1.0, output_bias=0.0, output_scale=1.0, skip_in=()):
        super().__init__()
        self.squeeze_out = squeeze_out
        dims = [d_in] + [d_hidden] * n_layers + [d_out]
        self.embedview_fn = None
        if multires_view > 0:
            self.embedview_fn, ch = get_embedder(multires_view, input_dims=1)
            dims[0] += ch - 1
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                dims[l] += dims[0]
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            setattr(self, "lin" + str(l), WNLinear(dims[l], out_dim, weight_norm))
        self.output_bias, self.output_scale, self.squeeze_out_scale = output_bias, output_scale, squeeze_out_scale

    def forward(self, distance, roughness, dot_n_wo, features=None):
        if self.embedview_fn is not None:
            dot_n_wo = self.embedview_fn(dot_n_wo)
        inp = torch.cat([distance, roughness, dot_n_wo] + ([features] if features is not None else []), dim=-1)
        x = _relu_mlp(self, inp)
        x = self.output_scale * (x + self.output_bias)
        if self.squeeze_out:
            return self.squeeze_out_scale * torch.sigmoid(x)
        return F.relu(x) * 2.0 * np.pi


class SingleVarianceNetwork(nn.Module):
    """inv_s = exp(10 * variance) (reference: src/network.py:370-379)."""

    def __init__(self, init_val):
        super().__init__()
        self.variance = nn.Parameter(torch.tensor(init_val))

    def forward(self, x):
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)


class DINONetwork(nn.Module):
    """3 -> 384 feature field (reference: src/network.py:382-486)."""

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(), multires=0, output_bias=0.0, output_scale=1.0,
                 weight_norm=True):
        super().__init__()
        dims = [d_in] + [d_hidden] * n_layers + [d_out]
        self.embed_fn = None
        if multires > 0:
            self.embed_fn, ch = get_embedder(multires)
            dims[0] += ch - 3
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        for l in range(self.num_layers - 1):
            if l in self.skip_in:
                dims[l] += dims[0]
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            setattr(self, "lin" + str(l), WNLinear(dims[l], out_dim, weight_norm))
        self.output_bias, self.output_scale = output_bias, output_scale

    def forward(self, points, normals=None, features=None, dino_feature_split=None, is_training=False):
        if self.embed_fn is not None:
            points = self.embed_fn(points)
        if normals is not None:
            parts = [points, normals, features] + ([dino_feature_split] if dino_feature_split is not None else [])
            inp = torch.cat(parts, dim=-1)
        else:
            inp = points
        x = _relu_mlp(self, inp, keep_graph=is_training)
        x = self.output_scale * (x + self.output_bias)
        return x if is_training else x.detach()

    def passback(self, points, feature, is_training=True):
        """d feature / d points summed over the feature channels (reference: src/network.py:469-486); `feature` must
        have been computed from `points` with is_training=True.  With is_training the result is differentiable once
        more (through the eager expression of the MLP: the fused kernels have no second-order path for 384 outputs)."""
        with torch.enable_grad():
            points.requires_grad_(True)
            (gradients,) = torch.autograd.grad(feature, points, torch.ones_like(feature), create_graph=is_training,
                                               retain_graph=is_training, only_inputs=True)
        if not is_training:
            return feature.detach(), gradients.detach()
        return feature, gradients
