"""The `render_fn` plug that render_surface.py hands to render_camera (reference: render_surface.py:40-176).

The reference defines render_fn / get_materials at module level with globals (args.is_metal,
args.smoothness_weight, ggx_renderer_modified, dino_dimension); `make_render_fn` binds those explicitly and
returns a callable with the reference's signature and result dictionary.
"""
import torch

dino_dimension = 384


def get_material_grads(value, points):
    """d value / d points for the smoothness loss (render_surface.py:40-48)."""
    return torch.autograd.grad(value, points, torch.ones_like(value), create_graph=True)[0]


def make_render_fn(ggx_renderer, is_metal=False, smoothness_weight=0.0):
    def get_materials(color_network_dict, points, normals, geo_f, dino_f, roughness_warmup=False, is_training=True):
        """diffuse(3) + specular(3) + roughness(1) (render_surface.py:51-86)."""
        nets = color_network_dict
        smooth = smoothness_weight > 0 and is_training and not roughness_warmup
        kd = nets["diffuse_albedo_network"](points, normals, -normals, geo_f, None).abs()[..., [2, 1, 0]]
        ks = nets["specular_albedo_network"](points, normals, None, geo_f, dino_f).abs()
        if not is_metal:
            ks = ks.mean(dim=-1, keepdim=True).expand_as(ks)
        if roughness_warmup:
            rough = torch.full((ks.shape[0], 1), 0.10, device=points.device)
        else:
            rough = nets["specular_roughness_network"](points, normals, None, geo_f, dino_f).abs() + 0.01
        if smooth:   # differentiates the material outputs w.r.t. the points once more (fused: pir_mlp_backward_tangent)
            ks_g, rough_g = get_material_grads(ks, points), get_material_grads(rough, points)
        else:
            ks_g, rough_g = torch.zeros_like(ks), torch.zeros_like(rough[..., 0:1])
        return kd, ks, rough, ks_g, rough_g

    def render_fn(interior_mask, color_network_dict, ray_o, ray_d, points, normals, geo_f, dino_f, is_training=True,
                  light=None, lightdir=None, distance=None, roughness_warmup=False, EDirectRadiance=True):
        n = list(interior_mask.shape)
        dev = interior_mask.device
        names3 = ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo", "specular_grads",
                  "roughness_grads", "normal")
        out = {k: torch.zeros(n + [3], dtype=torch.float32, device=dev) for k in names3}
        out["specular_roughness"] = torch.zeros(n, dtype=torch.float32, device=dev)
        from .renderer.raytracer import mask_indices
        flat_idx = mask_indices(interior_mask) if interior_mask.dim() == 1 else None   # shared with the caller's gather
        idx = (flat_idx,) if flat_idx is not None else interior_mask.nonzero(as_tuple=True)
        any_hit = idx[0].numel() > 0
        if any_hit:
            from .shade_hip import UnitNormals
            if isinstance(normals, UnitNormals):   # already n / (|n| + 1e-10), from the fused shading-input kernel
                normals = normals.as_subclass(torch.Tensor)
            else:
                normals = normals / (normals.norm(dim=-1, keepdim=True) + 1e-10)
            kd, ks, rough, ks_g, rough_g = get_materials(color_network_dict, points, normals, geo_f, dino_f,
                                                         is_training=is_training, roughness_warmup=roughness_warmup)
            if light is None:
                light = color_network_dict["point_light_network"]()
            if lightdir is None:
                lightdir = ray_d  # co-located light
            if distance is None:
                distance = (points - ray_o).norm(dim=-1, keepdim=True)
            sh = ggx_renderer(light, distance, normals, -lightdir, -ray_d, kd, ks, rough,
                              EDirectRadiance=EDirectRadiance)
            put = lambda key, v: out.__setitem__(key, out[key].index_put(idx, v))  # noqa: E731
            put("color", sh["rgb"])
            put("diffuse_color", sh["diffuse_rgb"])
            put("specular_color", sh["specular_rgb"])
            put("diffuse_albedo", kd)
            put("specular_albedo", ks)
            put("specular_grads", ks_g)
            put("roughness_grads", rough_g.expand(-1, 3) if rough_g.shape[-1] == 1 else rough_g)
            put("specular_roughness", rough.squeeze(-1))
            put("normal", normals)
        if dino_f is not None:
            dino = torch.zeros(n + [dino_dimension], dtype=torch.float32, device=dev)
            if any_hit:
                dino = dino.index_put(idx, dino_f)
            out["dino_feature"] = dino
        return out

    render_fn.get_materials = get_materials
    render_fn.accepts_unit_normals = True
    return render_fn
