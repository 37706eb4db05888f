"""Shared body of the multi-step trajectory case (tests/golden/train_traj.npz: FIVE consecutive iterations of the
reference's training loop, render_surface.py:545-699, global_step 11998..12002 with render_visibility_step = 12000, so
that the optimizer schedule changes phase in the middle -- every optimizer steps in the first two iterations, only
the SDF and the point light in the last three).  What one step cannot show: Adam state carried across steps, stale
packed weights after an optimizer step, None-gradient handling of the networks that stop being stepped, the
tracer status accumulated across steps."""
import numpy as np
import torch

from conftest import load_golden, state_from_npz

WATCH = {"sdf.lin0.weight_v": ("sdf", "lin0.weight_v", 1e-5), "sdf.lin4.weight_g": ("sdf", "lin4.weight_g", 1e-5),
         "sdf.lin8.bias": ("sdf", "lin8.bias", 1e-5), "sdf.lin8.weight_v": ("sdf", "lin8.weight_v", 1e-5),
         "diffuse.lin0.bias": ("diffuse_albedo_network", "lin0.bias", 1e-4),
         "diffuse.lin8.weight_v": ("diffuse_albedo_network", "lin8.weight_v", 1e-4),
         "specular.lin2.weight_g": ("specular_albedo_network", "lin2.weight_g", 1e-4),
         "roughness.lin4.bias": ("specular_roughness_network", "lin4.bias", 1e-4),
         "blend.lin0.bias": ("blend_network", "lin0.bias", 1e-4)}
MATERIAL = ("diffuse.lin0.bias", "diffuse.lin8.weight_v", "specular.lin2.weight_g", "roughness.lin4.bias", "blend.lin0.bias")


def build_state(device):
    import train_step_util
    from pir_amd import train
    d = load_golden("train_traj.npz")
    _, state = train_step_util.build_state(device)      # same networks / cameras / images as train_step.npz ...
    assert np.array_equal(state.gt_images.numpy(), d["gt_images"])
    state.sdf_network.load_state_dict(state_from_npz(d, "sdf."))
    state.args = train.TrainConfig(num_iters=12003, patch_size=32, render_visibility=True, smoothness_weight=1e-4,
                                   render_visibility_step=int(d["render_visibility_step"]))   # ... other schedule
    return d, state


def _param(state, key):
    net, name, _ = WATCH[key]
    mod = state.sdf_network if net == "sdf" else state.color_network_dict[net]
    return dict(mod.named_parameters())[name]


def run(device, loss_tol, flip_budget=3, exact_masks=True, step_tol=(0.02, 0.25)):
    from pir_amd import train
    d, state = build_state(device)
    K = int(d["n_steps"])
    np.random.seed(int(d["np_seed"]))
    pl = state.color_network_dict["point_light_network"]
    npy = lambda t: t.detach().cpu().numpy()  # noqa: E731
    for key in WATCH:
        assert np.array_equal(npy(_param(state, key)), d["before." + key]), key
    comparable = True
    report = []
    mine = {key: npy(_param(state, key)).copy() for key in MATERIAL}
    for i in range(K):
        step = int(d["first_step"]) + i
        out = train.training_step(state, step, eik_points=torch.from_numpy(d["eik_points"][i]).to(device))
        assert out["idx"] == int(d["steps.idx"][i]) and out["render_shadow"] == bool(d["steps.render_shadow"][i])
        assert np.array_equal(npy(out["camera_crop"].K), d["steps.K_crop"][i])
        dm = (npy(out["convergent_mask"]) != d[f"mask.{i}"]).sum() + (npy(out["results"]["edge_mask"]) != d[f"edge_mask.{i}"]).sum()
        assert dm <= (0 if exact_masks else flip_budget), (i, dm)
        comparable = comparable and dm == 0
        if comparable:       # a flipped pixel changes the image losses of this and (through the weights) later steps
            for k in ("loss", "img_loss", "img_l1_loss", "img_l2_loss", "img_ssim_loss", "eik_loss", "smoothness_loss"):
                ref = float(d["steps." + k][i])
                assert abs(out[k] - ref) <= loss_tol * abs(ref) + 1e-9, (i, k, out[k], ref)
            report.append((step, out["loss"], float(d["steps.loss"][i])))
        # optimizer schedule (render_surface.py:687-699): material networks step only while no shadows are rendered
        for key in MATERIAL:
            prev = d[f"after{i - 1}." + key] if i else d["before." + key]
            now = npy(_param(state, key)).copy()
            moved = not np.array_equal(now, mine[key])
            mine[key] = now
            ref_moved = not np.array_equal(d[f"after{i}." + key], prev)
            assert not (ref_moved and bool(d["steps.render_shadow"][i]))     # (a zero gradient moves nothing either)
            if comparable or not ref_moved:
                assert moved == ref_moved, (i, key, moved, ref_moved)
    # parameters after the five steps: Adam moves every element by about lr per step (lr * g / sqrt(v)), so the
    # comparison is on the displacement, in units of lr; elements whose gradient is rounding noise may step the other way
    worst = {}
    if comparable:
        for key, (_, _, lr) in WATCH.items():
            got = npy(_param(state, key)).astype(np.float64) - d["before." + key]
            ref = d[f"after{K - 1}." + key].astype(np.float64) - d["before." + key]
            err = np.abs(got - ref) / lr
            worst[key] = (float(np.median(err)), float(np.quantile(err, 0.99)), float(err.max()))
            assert np.median(err) < step_tol[0] and np.quantile(err, 0.95) < step_tol[1], (key, worst[key])
        for key, ref, lr in (("light", d[f"after{K - 1}.light"], 2e-5), ("offset", d[f"after{K - 1}.offset"], 1e-4)):
            got = npy(pl.light if key == "light" else pl.light_offset.weight)
            assert np.abs(got - ref).max() < 0.05 * lr * K + 1e-6 * abs(float(np.abs(ref).max())), (key, got, ref)
    state.raytracer.check_status()
    return {"comparable": comparable, "losses": report, "param_err_in_lr_units(median, q99, max)": worst}
