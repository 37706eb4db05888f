"""Synthetic scene of SURVEY.md section 8(d): random-init networks with the hyper-parameters of the reference's
configs/model_config.json, a camera at distance 3 looking at the origin, the Mitsuba default light offset."""
import math

import torch

from .network import SDFNetwork, RenderingNetwork, PointLightNetwork, DINONetwork, BlendNetwork
from .renderer.raytracer import Camera

MODEL_CONFIG = {
    "sdf_network": dict(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                        geometric_init=True, weight_norm=True),
    "diffuse_albedo_network": dict(d_in=9, d_out=3, d_feature=256, d_hidden=256, n_layers=8, multires=10,
                                   multires_view=4, mode="idr", squeeze_out=True, skip_in=[4]),
    "specular_albedo_network": dict(d_in=262, d_out=3, d_feature=384, d_hidden=256, n_layers=4, multires=6,
                                    multires_view=-1, mode="no_view_dir", squeeze_out=False, output_bias=0.4,
                                    output_scale=0.1),
    "specular_roughness_network": dict(d_in=262, d_out=1, d_feature=384, d_hidden=256, n_layers=4, multires=6,
                                       multires_view=-1, mode="no_view_dir", squeeze_out=False, output_bias=0.1,
                                       output_scale=0.1),
    "dino_network": dict(d_in=3, d_out=384, d_hidden=256, n_layers=4, skip_in=[], multires=6, output_bias=0.0,
                         output_scale=1.0),
    "blend_network": dict(d_in=3, d_out=1, d_hidden=128, n_layers=4, multires_view=6, squeeze_out=False),
    "lr": {"sdf_network": 1e-5, "blend_network": 1e-4, "diffuse_albedo_network": 1e-4, "specular_albedo_network": 1e-4,
           "specular_roughness_network": 1e-4},
}
LIGHT_OFFSET = (0.2, 0.1, 0.0)     # mitsuba/render_rgb_flash_mat.py:17
INV_S = math.exp(10 * 0.3)         # SingleVarianceNetwork(0.3), render_surface.py:200,241
LIGHT_INTENSITY = 15.0 * 3.0 * 3.0  # init_light_scale * dist^2, render_surface.py:243-245


def build_networks(seed=0, device="cpu"):
    """All networks initialised on the CPU generator (so every rank and the CPU baseline see the same weights)."""
    torch.manual_seed(seed)
    sdf = SDFNetwork(**MODEL_CONFIG["sdf_network"])
    mats = {k: RenderingNetwork(**MODEL_CONFIG[k]) for k in
            ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network")}
    mats["point_light_network"] = PointLightNetwork(enable_offset=True)
    mats["point_light_network"].set_light(LIGHT_INTENSITY)
    mats["point_light_network"].light_offset.weight.data = torch.tensor([list(LIGHT_OFFSET)])
    dino = DINONetwork(**MODEL_CONFIG["dino_network"])
    mats["blend_network"] = BlendNetwork(**MODEL_CONFIG["blend_network"])   # drawn last: earlier weights unchanged
    sdf = sdf.to(device)
    mats = {k: v.to(device) for k, v in mats.items()}
    return sdf, mats, dino.to(device)


def make_camera(W, H, view=0, n_views=1, focal_mul=2.0, device="cpu"):
    """K = [[f,0,W/2],[0,f,H/2]], W2C = rotation about y by 2 pi view / n_views, t_z = 3."""
    f = focal_mul * W
    K = torch.tensor([[f, 0, W / 2, 0], [0, f, H / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
    a = 2 * math.pi * view / max(1, n_views)
    c, s = math.cos(a), math.sin(a)
    W2C = torch.eye(4)
    W2C[:3, :3] = torch.tensor([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    W2C[2, 3] = 3.0
    return Camera(W, H, K.to(device), W2C.to(device))


def step_loss(results):
    """sum(color * vis) + eikonal term on the returned normals."""
    color = results["color"]
    if "vis" in results:
        color = color * results["vis"][..., None]
    m = results["convergent_mask"]
    return color.sum() + ((results["normal"][m].norm(dim=-1) - 1.0) ** 2).sum()
