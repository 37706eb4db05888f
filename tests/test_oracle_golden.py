"""Pins oracle/pir_oracle.py to the golden vectors captured from the reference (tools/gen_golden.py)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, state_from_npz, rel_err, masks_agree, mostly_close
from oracle import pir_oracle as O

TOL = 2e-5  # same torch CPU kernels on both sides; slack only for op-order differences


def _sdf_cfg(d):
    return dict(multires=int(d["cfg_multires"]), skip_in=tuple(int(v) for v in d["cfg_skip_in"]),
                scale=float(d["cfg_scale"]))


@pytest.mark.parametrize("name", ["sdf_small.npz", "sdf_small_bumpy.npz", "sdf_full.npz"])
def test_sdf_forward_and_get_all(name):
    d = load_golden(name)
    st = state_from_npz(d)
    params = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    layers = O.mlp_layers(params)
    kw = _sdf_cfg(d)
    x = torch.from_numpy(d["x"])
    assert rel_err(O.sdf_forward(layers, x, **kw).detach(), d["forward"]) < TOL
    y, f, n = O.sdf_get_all(layers, x.clone(), is_training=False, **kw)
    assert rel_err(y, d["sdf"]) < TOL and rel_err(f, d["feature"]) < TOL and rel_err(n, d["normal"]) < TOL
    xt = x.clone()
    y, f, n = O.sdf_get_all(layers, xt, is_training=True, **kw)
    loss = (y * torch.from_numpy(d["cs"])).sum() + (f * torch.from_numpy(d["cf"])).sum() + \
        (n * torch.from_numpy(d["cn"])).sum()
    loss.backward()
    assert abs(loss.item() - float(d["loss"])) < 1e-4 * abs(float(d["loss"])) + 1e-5
    assert rel_err(xt.grad, d["grad_x"]) < 1e-4
    for k, p in params.items():
        g = p.grad.numpy()
        ref = d["g." + k]
        if g.shape != ref.shape:
            g = g[::8]
        assert rel_err(g, ref) < 1e-4, k


def test_camera_and_sphere():
    d = load_golden("camera.npz")
    cam = O.CameraRays(int(d["W"]), int(d["H"]), torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"]))
    uv = cam.pixel_centers()
    assert np.array_equal(uv.numpy(), d["uv"])
    o, dd, nn = cam.rays(uv)
    assert rel_err(o, d["ray_o"]) < 1e-6 and rel_err(dd, d["ray_d"]) < 1e-6 and rel_err(nn, d["ray_d_norm"]) < 1e-6
    m, lo, hi = O.intersect_sphere(o.reshape(-1, 3), dd.reshape(-1, 3), 1.0)
    assert np.array_equal(m.numpy(), d["mask"])
    assert rel_err(lo, d["min_dis"]) < 1e-6 and rel_err(hi, d["max_dis"]) < 1e-6
    assert rel_err(cam.light_pos(torch.from_numpy(d["offset"])), d["light_pos"]) < 1e-6


@pytest.mark.parametrize("name,weights", [("trace_std64.npz", "sdf_full.npz"), ("trace_wide48.npz", "sdf_full.npz"),
                                          ("trace_bumpy64.npz", "sdf_full_bumpy_weights.npz"),
                                          ("trace_crop800.npz", "sdf_full_bumpy_weights.npz"),
                                          ("trace_vis32.npz", "sdf_full_bumpy_weights.npz")])
def test_tracer(name, weights):
    d = load_golden(name)
    layers = O.mlp_layers(state_from_npz(load_golden(weights)))
    cam = O.CameraRays(int(d["W"]), int(d["H"]), torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"]))
    tr = O.Tracer()
    with torch.no_grad():
        o, dd, _ = cam.rays(cam.pixel_centers())
        o, dd = o.reshape(-1, 3), dd.reshape(-1, 3)
        m, lo, hi = O.intersect_sphere(o, dd, 1.0)
        sdf = lambda x: O.sdf_forward(layers, x)[..., 0]  # noqa
        conv, unf, pts, val, acc = tr.sphere_trace(sdf, o, dd, lo, hi, m)
    # after the 16 sphere-tracing steps a ray at the threshold may sit on either side of it (the sampler then
    # finishes it): a few flips between "convergent" and "unfinished", one step (<= ~5e-5) of position
    n = conv.numel()
    assert (conv.numpy() != d["st_convergent"]).sum() <= max(1, 2e-3 * n)
    assert (unf.numpy() != d["st_unfinished"]).sum() <= max(1, 2e-3 * n)
    same = (conv.numpy() == d["st_convergent"]) & (unf.numpy() == d["st_unfinished"])
    e_pts, e_dis = np.abs(pts.numpy() - d["st_points"]).max(-1)[same], np.abs(acc.numpy() - d["st_dis"])[same]
    assert e_pts.max() < 1.1e-4 and e_dis.max() < 1.1e-4
    assert (e_pts > 1e-5).mean() < 0.02 and (e_dis > 1e-5).mean() < 0.02
    res = O.raytrace_camera(cam, layers, tr)
    assert masks_agree(res["convergent_mask"].numpy(), d["cam_mask"])
    both = res["convergent_mask"].numpy() & d["cam_mask"]
    e_depth = np.abs(res["depth"].numpy() - d["cam_depth"])[both]
    e_pts = np.abs(res["points"].numpy() - d["cam_points"]).max(-1)[both]
    assert e_depth.max() <= 1e-4 * d["cam_depth"].max() + 1e-5 and e_pts.max() < 3e-4     # 1e-4 relative (north_star)
    assert (e_depth > 1e-5).mean() < 0.03 and (e_pts > 1e-5).mean() < 0.03


@pytest.mark.parametrize("n_samples", [128, 256])
def test_visibility(n_samples):
    d = load_golden("visibility.npz")
    st = state_from_npz(load_golden("sdf_full_bumpy_weights.npz"))
    params = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    layers = O.mlp_layers(params)
    cam = O.CameraRays(32, 32, torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"]))
    off = torch.tensor([[0.2, 0.1, 0.0]], requires_grad=True)
    lp = cam.light_pos(off)
    pts = torch.from_numpy(d["points"])
    inv_s = torch.from_numpy(d["inv_s"])
    vis = O.light_visibility(lp, pts, layers, inv_s, n_samples=n_samples, is_training=True)
    assert rel_err(vis.detach(), d[f"vis{n_samples}"]) < TOL
    (vis * torch.from_numpy(d[f"c{n_samples}"])).sum().backward()
    assert rel_err(off.grad, d[f"grad_offset{n_samples}"]) < 1e-4
    for k, p in params.items():
        g = p.grad.numpy()
        ref = d[f"g{n_samples}." + k]
        if g.shape != ref.shape:
            g = g[::8]
        assert rel_err(g, ref) < 2e-4, k
    if n_samples == 128:
        v2 = O.light_visibility(torch.from_numpy(d["light_pos_side"]), pts, O.mlp_layers(st), inv_s, 128, False)
        assert rel_err(v2, d["vis_side128"]) < TOL


def _tables():
    import os
    from conftest import ROOT
    return O.load_ggx_tables(os.path.join(ROOT, "pir_amd", "renderer", "ggx"))


def test_ggx_golden_and_known_answers():
    d = load_golden("ggx.npz")
    tabs = _tables()
    names = ("distance", "normal", "lightdir", "viewdir", "kd", "ks", "alpha")
    for tag, edr in (("offset", True), ("coloc", True), ("noedr", False)):
        leaves = {n: torch.from_numpy(d[n]).clone().requires_grad_(True) for n in names}
        light = torch.tensor(float(d["light"]), requires_grad=True)
        ld = leaves["viewdir"] if tag == "coloc" else leaves["lightdir"]
        o = O.ggx_shade(light, leaves["distance"], leaves["normal"], ld, leaves["viewdir"], leaves["kd"], leaves["ks"],
                        leaves["alpha"], tabs, direct_radiance=edr)
        for k in ("diffuse_rgb", "specular_rgb", "rgb"):
            assert rel_err(o[k].detach(), d[f"{tag}.{k}"]) < TOL, (tag, k)
        (o["rgb"] * torch.from_numpy(d[f"{tag}.cw"])).sum().backward()
        assert rel_err(light.grad, d[f"{tag}.g_light"]) < 1e-4
        for n in names:
            if tag == "coloc" and n == "lightdir":
                continue
            g = leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(leaves[n])
            assert rel_err(g, d[f"{tag}.g_{n}"]) < 1e-4, (tag, n)
    o = O.ggx_shade(torch.from_numpy(d["rgblight.light"]), *[torch.from_numpy(d[n]) for n in names], tabs,
                    direct_radiance=False)
    assert rel_err(o["rgb"], d["rgblight.rgb"]) < TOL
    # known answers from SURVEY.md Appendix A (captured from the reference)
    one = lambda *v: torch.tensor([list(v)], dtype=torch.float32)  # noqa
    nrm = one(0, 0, 1)
    o = O.ggx_shade(torch.tensor(10.0), one(2.0), nrm, nrm, nrm, one(.5, .5, .5), one(.5, .5, .5), one(0.1), tabs)
    assert abs(o["diffuse_rgb"][0, 0].item() - 0.4032129943) < 1e-6
    assert abs(o["specular_rgb"][0, 0].item() - 0.3831406832) < 1e-6
    l = torch.nn.functional.normalize(one(0.3, 0.1, 1.0), dim=-1)
    v = torch.nn.functional.normalize(one(-0.2, 0.4, 1.0), dim=-1)
    o = O.ggx_shade(torch.tensor(10.0), one(2.0), nrm, l, v, one(.5, .5, .5), one(.5, .5, .5), one(0.3), tabs)
    assert abs(o["diffuse_rgb"][0, 0].item() - 0.3632915318) < 1e-6
    assert abs(o["specular_rgb"][0, 0].item() - 0.0172253586) < 1e-7


def test_ggx_table_bins_equal_the_reference_bit_for_bit():
    """The shader's index work on the CPU (tests/golden/indices.npz: the outputs of the reference's own torch.floor
    calls on the 4 096 points of ggx.npz): integer equality, no tolerance."""
    d, ix = load_golden("ggx.npz"), load_golden("indices.npz")
    tabs = _tables()
    lo, hi = 0.00001, 0.99999
    nrm, view, a = torch.from_numpy(d["normal"]), torch.from_numpy(d["viewdir"]), torch.from_numpy(d["alpha"]).clamp(min=1e-4)
    for tag, ld in (("offset", torch.from_numpy(d["lightdir"])), ("coloc", view)):
        c_o = (view * nrm).sum(-1, keepdim=True).clamp(lo, hi)
        c_i = (ld * nrm).sum(-1, keepdim=True).clamp(lo, hi)
        bi, bo, bd = O.ggx_table_bins(c_i, c_o, a, tabs[0].numel(), tabs[1].numel())
        got = {"idx_wi": bi, "idx_wo": bo, "idx_diff": bd}
        for k, v in got.items():
            assert np.array_equal(v.reshape(-1).numpy().astype(np.int32), ix[f"ggx.{tag}.{k}"]), (tag, k)


@pytest.mark.parametrize("name,weights", [("trace_std64", "sdf_full.npz"), ("trace_bumpy64", "sdf_full_bumpy_weights.npz"),
                                          ("trace_crop800", "sdf_full_bumpy_weights.npz")])
@pytest.mark.parametrize("iters", [0, 16])
def test_sampler_first_negative_index(name, weights, iters):
    """The dense sampler's index work (raytracer.py:192-197) of the restatement against the reference's torch.min
    output.  Without sphere tracing the sampler sees identical intervals: equal, up to samples whose |sdf| is fp32
    round-off.  After 16 sphere-tracing steps the yardstick is the reference against itself under another fp32
    evaluation order (1 thread / 997-row slices; 44 of 1 848 rays on the silhouette crop, 0 on the 64 x 64 views)."""
    d, ix = load_golden(name + ".npz"), load_golden("indices.npz")
    layers = O.mlp_layers(state_from_npz(load_golden(weights)))
    cam = O.CameraRays(int(d["W"]), int(d["H"]), torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"]))
    tr = O.Tracer(sphere_tracing_iters=iters)
    with torch.no_grad():
        o, dd, _ = cam.rays(cam.pixel_centers())
        o, dd = o.reshape(-1, 3), dd.reshape(-1, 3)
        sdf = lambda x: O.sdf_forward(layers, x)[..., 0]  # noqa
        lo, hi, wm = torch.from_numpy(d["min_dis"]), torch.from_numpy(d["max_dis"]), torch.from_numpy(d["work_mask"])
        _, unf, *_ = tr.sphere_trace(sdf, o, dd, lo, hi, wm)
        tr.trace(sdf, o, dd, lo, hi, wm)
    key = name + (".it0" if iters == 0 else "")
    mine = dict(zip(np.nonzero(unf.numpy())[0].tolist(),
                    torch.where(tr.last_min_negative, tr.last_min_idx, torch.full_like(tr.last_min_idx, -1)).tolist()))

    def first_negative(k):
        return dict(zip(ix[k + ".sampler_rays"].tolist(), np.where(ix[k + ".min_val_negative"], ix[k + ".min_idx"], -1).tolist()))
    ref = first_negative(key)
    assert len(set(mine) ^ set(ref)) <= max(2, len(ref) // 500)
    differ = [r for r in mine if r in ref and mine[r] != ref[r]]
    if iters == 0:
        assert len(differ) <= max(2, len(ref) // 200), len(differ)
        for r in differ:       # the earlier of the two indices is the disputed sample: its value must be round-off
            k = min(v for v in (mine[r], ref[r]) if v >= 0)
            z = lo[r] + torch.linspace(0, 1, tr.n_steps)[k] * (hi[r] - lo[r])
            assert abs(float(sdf((o[r] + dd[r] * z)[None])[0])) < 1e-5
    else:
        own = max(sum(ref[r] != v for r, v in first_negative(f"{key}.{alt}").items() if r in ref)
                  for alt in ("alt_t1", "alt_rows997"))
        assert len(differ) <= 2 * own, (len(differ), own)


def test_reparam():
    d = load_golden("reparam.npz")
    s = torch.from_numpy(d["sdf"]).clone().requires_grad_(True)
    out = O.reparam_points(torch.from_numpy(d["points"]), torch.from_numpy(d["grads"]), torch.from_numpy(d["dirs"]), s)
    assert np.array_equal(out.detach().numpy(), d["out"])
    (out * torch.from_numpy(d["cw"])).sum().backward()
    assert rel_err(s.grad, d["g_sdf"]) < 1e-6


MAT_CFG = dict(
    diffuse_albedo_network=dict(multires=10, multires_view=4, mode="idr", squeeze_out=True, skip_in=[4]),
    specular_albedo_network=dict(multires=6, multires_view=-1, mode="no_view_dir", squeeze_out=False,
                                 output_bias=0.4, output_scale=0.1),
    specular_roughness_network=dict(multires=6, multires_view=-1, mode="no_view_dir", squeeze_out=False,
                                    output_bias=0.1, output_scale=0.1),
)
DINO_CFG = dict(multires=6, output_bias=0.0, output_scale=1.0)


def mats_from(d, requires_grad=False):
    mats, params = {}, {}
    for nm, cfg in MAT_CFG.items():
        st = state_from_npz(d, f"mat.{nm}.")
        if requires_grad:
            st = {k: v.clone().requires_grad_(True) for k, v in st.items()}
        params[nm] = st
        mats[nm] = dict(layers=O.mlp_layers(st), cfg=cfg)
    dino = dict(layers=O.mlp_layers(state_from_npz(d, "mat.dino_network.")), cfg=DINO_CFG)
    return mats, dino, params


def test_render_fn():
    d = load_golden("render_fn.npz")
    mats, _, _ = mats_from(d)
    tabs = _tables()
    t = lambda k: torch.from_numpy(d[k])  # noqa
    light = torch.tensor(float(d["mat.point_light_network.light"]))
    for tag, kw in (("coloc", {}), ("offset", dict(lightdir=t("offset.lightdir"), distance=t("offset.distance"))),
                    ("warmup", dict(roughness_warmup=True))):
        with torch.no_grad():
            r = O.render_fn(t("interior_mask"), mats, tabs, t("ray_o"), t("ray_d"), t("points"), t("normals"),
                            t("geo_f"), t("dino_f"), light, **kw)
        for k in ("color", "diffuse_color", "specular_color", "diffuse_albedo", "specular_albedo",
                  "specular_roughness", "normal", "specular_grads", "roughness_grads"):
            assert rel_err(r[k], d[f"{tag}.{k}"]) < TOL, (tag, k)


@pytest.mark.parametrize("tag", ["coloc", "offset"])
def test_end_to_end_tile(tag):
    d = load_golden("e2e_tile.npz")
    rf = load_golden("render_fn.npz")
    st = state_from_npz(load_golden("sdf_full_bumpy_weights.npz"))
    sdf_params = {k: v.clone().requires_grad_(True) for k, v in st.items()}
    layers = O.mlp_layers(sdf_params)
    mats, dino, mat_params = mats_from(rf, requires_grad=True)
    cam = O.CameraRays(32, 32, torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"]))
    light = torch.tensor(135.0, requires_grad=True)
    off = torch.tensor([[0.2, 0.1, 0.0]], requires_grad=True) if tag == "offset" else None
    res = O.render_view(cam, layers, mats, dino, _tables(), light, light_offset=off,
                        inv_s=torch.from_numpy(d["inv_s"]), is_training=True)
    assert masks_agree(res["convergent_mask"].numpy(), d[f"{tag}.mask"])
    both = res["convergent_mask"].numpy() & d[f"{tag}.mask"]
    # TOL on 97 % of the hit pixels; the few whose ray stopped a step earlier/later at the tracer threshold on this
    # host's GEMM summation order (see conftest.mostly_close) are bounded at 2e-3
    assert mostly_close(res["color"].detach(), d[f"{tag}.color"], soft=TOL, sel=both)
    assert mostly_close(res["normal"].detach(), d[f"{tag}.normal"], soft=TOL, sel=both)
    assert np.abs(res["depth"].numpy() - d[f"{tag}.depth"])[both].max() < 2.5e-4
    if not np.array_equal(res["convergent_mask"].numpy(), d[f"{tag}.mask"]):
        return      # a pixel flipped at the threshold: sums and gradients are not comparable one to one
    loss = O.step_loss(res)
    assert abs(loss.item() - float(d[f"{tag}.loss"])) < 1e-4 * abs(float(d[f"{tag}.loss"]))
    loss.backward()
    if tag == "offset":
        assert mostly_close(res["vis"].detach(), d["offset.vis"], soft=TOL, sel=both)
        assert rel_err(off.grad, d["offset.g_offset"]) < 2e-4
    assert rel_err(light.grad, d[f"{tag}.g_light"]) < 1e-4
    for k, p in sdf_params.items():
        g, ref = p.grad.numpy(), d[f"{tag}.g.sdf.{k}"]
        if g.shape != ref.shape:
            g = g[::8]
        assert rel_err(g, ref) < 3e-4, k
    for nm, ps in mat_params.items():
        for k, p in ps.items():
            g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
            ref = d[f"{tag}.g.{nm}.{k}"]
            if g.shape != ref.shape:
                g = g[::8]
            assert rel_err(g, ref) < 1e-3, (nm, k)


BLEND_CFG = dict(multires_view=6, squeeze_out=False)


def test_next_rows_edges_and_indirect():
    """locate_edge_points, render_indirect_lighting and render_edge_pixels of the reference on a 48x48 view of a
    strongly perturbed sphere (24 edge pixels, 70 pixels with a secondary hit), forward and backward."""
    d = load_golden("next_rows.npz")
    rf = load_golden("render_fn.npz")
    sdf_params = {k: v.clone().requires_grad_(True) for k, v in state_from_npz(d, "sdf.").items()}
    layers = O.mlp_layers(sdf_params)
    mats, dino, mat_params = mats_from(rf, requires_grad=True)
    bst = {k: v.clone().requires_grad_(True) for k, v in state_from_npz(d, "mat.blend_network.").items()}
    mats["blend_network"] = dict(layers=O.mlp_layers(bst), cfg=BLEND_CFG)
    mat_params["blend_network"] = bst
    cam = O.CameraRays(48, 48, torch.from_numpy(d["K"]), torch.from_numpy(d["W2C"]))
    tabs, tracer = _tables(), O.Tracer()
    light = torch.tensor(135.0, requires_grad=True)
    off = torch.tensor([[0.2, 0.1, 0.0]], requires_grad=True)
    inv_s = torch.from_numpy(d["inv_s"])

    res = O.raytrace_camera(cam, layers, tracer)
    assert np.array_equal(res["convergent_mask"].numpy(), d["trace.mask"])     # (no ray of this view is marginal)
    # the Sobel restatement (kornia's published algorithm, unpinned) reproduces the mask stored with the fixture
    edge_in = (O.sobel_magnitude(res["depth"]) > 1e-2) & res["convergent_mask"]
    assert np.array_equal(edge_in.numpy(), d["depth_edge_mask"])
    edge = O.locate_edge_points(cam, res["points"], layers, edge_in)
    assert np.array_equal(edge["edge_mask"].numpy(), d["edge.edge_mask"])
    assert np.array_equal(edge["edge_pixel_idx"].numpy(), d["edge.edge_pixel_idx"])
    assert np.abs(edge["edge_points"].numpy() - d["edge.edge_points"]).max() < 1e-6
    assert np.abs(edge["edge_uv"].numpy() - d["edge.edge_uv"]).max() < 1e-3
    res.update(edge)
    res["convergent_mask"] = res["convergent_mask"] & ~res["edge_mask"]

    # visibility over hits + edge pixels, then direct shading
    light_pos = cam.light_pos(off)
    m_all = (res["convergent_mask"] | res["edge_mask"]).reshape(-1)
    vis = torch.zeros(48 * 48).index_put(
        (m_all.nonzero()[:, 0],),
        O.light_visibility(light_pos, res["points"].reshape(-1, 3)[m_all], layers, inv_s, 128, True))
    res["vis"] = vis.reshape(48, 48)
    out = O.shade_traced(res, layers, mats, dino, tabs, light, light_pos=light_pos, is_training=True)
    for k in ("color", "normal"):
        res[k] = out[k].reshape(48, 48, 3)
    hit = d["final.mask"] | d["edge.edge_mask"]
    # see conftest.mostly_close: TOL on 97 % of the pixels, 2e-3 on the few that sit at the tracer threshold
    assert mostly_close(res["color"].detach(), d["direct.color"], soft=TOL, sel=hit)
    ic, ib = O.render_indirect(res, layers, mats, dino, tabs, tracer, light, is_training=True)
    assert np.array_equal((ic.detach().numpy().sum(-1) != 0), (d["indirect_color"].sum(-1) != 0))
    lit = d["indirect_color"].sum(-1) != 0      # 70 pixels; the second bounce amplifies a shifted first hit
    e_ic = (np.abs(ic.detach().numpy() - d["indirect_color"]) / np.abs(d["indirect_color"]).max())[lit]
    # a second trace starts from the first hit along the mirrored direction, so it inherits the first hit's ~1e-4
    # sensitivity twice over: TOL on the median, 2e-4 on 97 %, 2e-3 everywhere
    assert np.median(e_ic) < TOL and mostly_close(ic.detach(), d["indirect_color"], soft=2e-4, sel=lit)
    assert mostly_close(ib.detach(), d["indirect_blend_coef"], soft=TOL, sel=lit)
    O.render_edge_pixels(res, cam, layers, mats, dino, tabs, tracer, light, light_pos=light_pos, is_training=True)
    assert mostly_close(res["color"].detach(), d["final.color"], soft=TOL, sel=hit)
    assert mostly_close(res["normal"].detach(), d["final.normal"], soft=TOL, sel=hit)
    e_pts = np.abs(res["points"].numpy() - d["final.points"]).max(-1)
    assert e_pts.max() < 3e-4 and (e_pts > 1e-5).mean() < 0.03
    assert mostly_close(res["edge_pos_neg_normal"].detach(), d["edge_pos_neg_normal"], soft=1e-4, q=0.9)

    loss = (res["color"] * res["vis"][..., None] + ic * ib).sum()
    assert abs(loss.item() - float(d["loss"])) < 1e-4 * abs(float(d["loss"]))
    loss.backward()
    assert rel_err(light.grad, d["g_light"]) < 1e-4
    assert rel_err(off.grad, d["g_offset"]) < 2e-4
    for k, p in sdf_params.items():
        g, ref = p.grad.numpy(), d[f"g.sdf.{k}"]
        assert rel_err(g[::8] if g.shape != ref.shape else g, ref) < 3e-4, k
    for nm, ps in mat_params.items():
        for k, p in ps.items():
            g = p.grad.numpy() if p.grad is not None else np.zeros(p.shape, np.float32)
            ref = d[f"g.{nm}.{k}"]
            assert rel_err(g[::8] if g.shape != ref.shape else g, ref) < 3e-4, (nm, k)
