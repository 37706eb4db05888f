#!/usr/bin/env python3
"""Full-size golden vectors (round 2): runs the *reference* PIR modules on CPU, in the build container only.

  trace_full800.npz   raytrace_camera of the reference on the 800 x 800 standard view (bumpy full-size SDF of
                      sdf_full_bumpy_weights.npz): the hit mask of all 640 000 pixels (bit-packed) and depth /
                      hit point / sdf of a seeded 50 000-pixel subsample.
  shade_points.npz    the shading half of the path with the tracer taken OUT of the loop: 2048 hit points of that
                      trace (ray_o, ray_d, points) pushed through the reference's light_visibility(128),
                      render_normal_and_color (get_all + reparam_points + render_fn with the FULL-SIZE material
                      networks + GGX, offset light) and the loss
                      sum(color * vis * cw) + eikonal, with every output and every parameter gradient -- once in fp32
                      (`ref32.*`, the reference as it runs) and once in fp64 (`ref64.*`, the same modules under
                      torch.set_default_dtype(float64); the hard-coded float32 result buffers of render_fn are widened
                      for this run only).  The fp64 run adjudicates tolerances: a kernel is as good as the reference
                      when it is no farther from fp64 than the fp32 reference is.

Same import recipe as tools/gen_golden.py (stubbed kornia/cv2/icecream/turtle, render_fn from the text of
render_surface.py:40-176).  Usage:  python tools/gen_golden_full800.py [--only trace|shade]
"""
import argparse
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as GG  # noqa: E402
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import materials_util as MU  # noqa: E402

GOLDEN = os.path.join(HERE, "..", "tests", "golden")
N_SUB = 50000
N_SHADE = 2048


def load_state(prefix, d):
    return {k[len(prefix):]: torch.from_numpy(d[k]) for k in d.files if k.startswith(prefix)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    ap.add_argument("--res", type=int, default=800)
    args = ap.parse_args()
    GG._install_stubs()
    sys.path.insert(0, args.ref)
    from src.network import SDFNetwork, RenderingNetwork, PointLightNetwork, DINONetwork  # noqa
    from src.renderer.renderer_ggx import GGXColocatedRenderer  # noqa
    import src.renderer.raytracer as rt  # noqa
    torch.set_num_threads(8)

    wts = np.load(os.path.join(GOLDEN, "sdf_full_bumpy_weights.npz"))
    net = SDFNetwork(**GG.SDF_FULL)
    net.load_state_dict(load_state("w.", wts))
    res = args.res
    cam = GG.standard_camera(rt.Camera, res, res)
    trace_path = os.path.join(GOLDEN, "trace_full800.npz")

    if args.only in ("", "trace"):
        t0 = time.time()
        with torch.no_grad():
            r = rt.raytrace_camera(cam, net, rt.RayTracer(), fill_holes=False, detect_edges=False)
        print(f"reference trace of {res}x{res}: {time.time() - t0:.0f} s, hits {int(r['convergent_mask'].sum())}", flush=True)
        mask = r["convergent_mask"].reshape(-1).numpy()
        rs = np.random.RandomState(800)
        sub = np.sort(rs.choice(res * res, N_SUB, replace=False)).astype(np.int32)
        np.savez_compressed(trace_path, K=GG._np(cam.K), W2C=GG._np(cam.W2C), W=res, H=res,
                            mask_bits=np.packbits(mask), n_hits=np.int64(mask.sum()), sub_idx=sub,
                            sub_depth=r["depth"].reshape(-1).numpy()[sub],
                            sub_points=r["points"].reshape(-1, 3).numpy()[sub],
                            sub_sdf=r["sdf"].reshape(-1).numpy()[sub],
                            sub_ray_d=r["ray_d"].reshape(-1, 3).numpy()[sub],
                            sub_ray_d_norm=r["ray_d_norm"].reshape(-1).numpy()[sub],
                            ray_o=r["ray_o"].reshape(-1, 3).numpy()[0])

    if args.only in ("", "shade", "noise"):
        tr = np.load(trace_path)
        mask = np.unpackbits(tr["mask_bits"])[:res * res].astype(bool)
        sub = tr["sub_idx"]
        hit_sub = np.nonzero(mask[sub])[0]
        rs = np.random.RandomState(801)
        pick = np.sort(rs.choice(hit_sub, N_SHADE, replace=False))
        pts32 = torch.from_numpy(tr["sub_points"][pick])
        rayd32 = torch.from_numpy(tr["sub_ray_d"][pick])
        rayo32 = torch.from_numpy(tr["ray_o"]).reshape(1, 3).expand(N_SHADE, 3).contiguous()
        cw32 = torch.from_numpy(rs.rand(N_SHADE, 3).astype(np.float32) + 0.5)
        import json
        cfg = json.load(open(os.path.join(args.ref, "configs", "model_config.json")))
        with open(os.path.join(args.ref, "render_surface.py")) as fh:
            text = "\n".join(fh.read().split("\n")[39:176])
        out = dict(points=pts32.numpy(), ray_d=rayd32.numpy(), ray_o=rayo32.numpy(), cw=cw32.numpy(),
                   pixel_idx=sub[pick], light_offset=np.array([[0.2, 0.1, 0.0]], np.float32),
                   inv_s=np.float32(np.exp(3.0)))

        # "noise": the fp32 reference re-run under other CPU thread counts (another GEMM blocking = another fp32
        # rounding of the same code); per tensor, the largest element-wise distance of those runs from the fp64 run
        # goes to shade_points_noise.npz -- the reference's own fp32 spread on this fixture (ReLU / clamp kinks of the
        # material networks flip under round-off of the normals that feed them), the yardstick of the GPU test
        runs = [("ref32", torch.float32, 8), ("ref64", torch.float64, 8)] if args.only != "noise" else \
            [(f"alt{t}", torch.float32, t) for t in (6, 4, 3, 2)] + [("flipk", torch.float32, 8), ("split2", torch.float32, 8),
                                                                    ("exact", torch.float32, 8)]
        real_linear = F.linear

        def flipk(x, w, b=None):            # K summed in reverse order
            return real_linear(x.flip(-1), w.flip(-1), b)

        def split2(x, w, b=None):           # two half-K products added
            h = x.shape[-1] // 2
            y = real_linear(x[..., :h], w[:, :h]) + real_linear(x[..., h:], w[:, h:])
            return y if b is None else y + b

        def exact(x, w, b=None):            # accumulated in double, rounded once per output
            if x.dtype != torch.float32:
                return real_linear(x, w, b)
            return real_linear(x.double(), w.double(), None if b is None else b.double()).float()
        for tag, dtype, threads in runs:
            torch.nn.functional.linear = {"flipk": flipk, "split2": split2, "exact": exact}.get(tag, real_linear)
            torch.set_num_threads(threads)
            torch.set_default_dtype(dtype)
            src = text if dtype == torch.float32 else text.replace("dtype=torch.float32", "dtype=torch.float64")
            ns = dict(torch=torch, nn=nn, F=F, np=np, dino_dimension=384,
                      args=Namespace(is_metal=False, smoothness_weight=0.0), ggx_renderer_modified=GGXColocatedRenderer())
            exec(src, ns)
            render_fn = ns["render_fn"]
            sdf_net = SDFNetwork(**GG.SDF_FULL)
            sdf_net.load_state_dict(load_state("w.", wts))
            # full-size material / DINO networks (configs/model_config.json); weights regenerated from numpy's
            # RandomState by tests/materials_util.fill_state (seeds 100..103, as in materials_full.npz)
            cnd = {}
            for i, nm in enumerate(MU.NAMES[:3]):
                cnd[nm] = RenderingNetwork(**cfg[nm])
                cnd[nm].load_state_dict(MU.fill_state(cnd[nm].state_dict(), seed=100 + i))
            MAT = {k: None for k in MU.NAMES[:3]}
            pl = PointLightNetwork(enable_offset=True)
            pl.set_light(135.0)
            pl.light_offset.weight.data = torch.tensor([[0.2, 0.1, 0.0]])
            cnd["point_light_network"] = pl
            dino = DINONetwork(**cfg["dino_network"])
            dino.load_state_dict(MU.fill_state(dino.state_dict(), seed=103))
            pts, rayd, rayo, cw = pts32.to(dtype), rayd32.to(dtype), rayo32.to(dtype), cw32.to(dtype)
            inv_s = torch.exp(torch.tensor(0.3) * 10.0).reshape(1, 1)
            light_pos = cam.get_light_pos(pl.offset) if dtype == torch.float32 else \
                torch.matmul(pl.offset, cam.C2W[:3, :3].double().transpose(1, 0)) + cam.C2W[:3, 3].double().unsqueeze(0)
            results = {"convergent_mask": torch.ones(N_SHADE, dtype=torch.bool), "points": pts.clone(),
                       "ray_d": rayd, "ray_o": rayo, "edge_mask": torch.zeros(N_SHADE, dtype=torch.bool)}
            t0 = time.time()
            # light_visibility directly (render_light_visibility only scatters it into a float32 image buffer)
            vis = rt.light_visibility(light_pos, pts.clone(), sdf_net, inv_s, n_samples=128, is_training=True)
            rt.render_normal_and_color(results, sdf_net, cnd, dino, render_fn, is_training=True, max_num_pts=320000,
                                       light_pos=light_pos)
            color, normal = results["color"], results["normal"]
            loss = (color * vis[..., None] * cw).sum() + ((normal.norm(dim=-1) - 1.0) ** 2).sum()
            loss.backward()
            print(f"{tag}: {time.time() - t0:.0f} s, loss {loss.item():.9g}", flush=True)
            f64 = lambda t: t.detach().double().numpy().copy() if dtype == torch.float64 else GG._np(t)  # noqa: E731
            out.update({f"{tag}.vis": f64(vis), f"{tag}.color": f64(color), f"{tag}.normal": f64(normal),
                        f"{tag}.diffuse_albedo": f64(results["diffuse_albedo"]),
                        f"{tag}.specular_albedo": f64(results["specular_albedo"]),
                        f"{tag}.specular_roughness": f64(results["specular_roughness"]),
                        f"{tag}.loss": np.float64(loss.item()), f"{tag}.g_light": f64(pl.light.grad),
                        f"{tag}.g_offset": f64(pl.light_offset.weight.grad)})

            def sub_grads(named):
                o = {}
                for n, p in named:
                    g = f64(p.grad if p.grad is not None else torch.zeros_like(p))
                    o[n] = g[::8] if g.ndim == 2 and g.shape[0] * g.shape[1] > 40000 else g
                return o
            out.update({f"{tag}.g.sdf." + k: v for k, v in sub_grads(sdf_net.named_parameters()).items()})
            for nm in MAT:
                out.update({f"{tag}.g.{nm}." + k: v for k, v in sub_grads(cnd[nm].named_parameters()).items()})
        torch.set_default_dtype(torch.float32)
        torch.set_num_threads(8)
        torch.nn.functional.linear = real_linear
        if args.only == "noise":
            sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
            from full800_util import elem_err
            fx = np.load(os.path.join(GOLDEN, "shade_points.npz"))
            noise = {}
            for k in fx.files:
                if not k.startswith("ref64."):
                    continue
                name = k[len("ref64."):]
                floor = np.abs(fx[k].astype(np.float64)).mean() + 1e-300
                errs = [elem_err(out[f"{tag}.{name}"], fx[k], floor) for tag, _, _ in runs]
                noise["noise." + name] = np.float64(max(errs))
                own = elem_err(fx["ref32." + name], fx[k], floor)
                if max(errs) > 2 * own and max(errs) > 1e-5:
                    print(f"  {name}: fp32 reference vs fp64 {own:.2e}, other thread counts {[f'{e:.2e}' for e in errs]}")
            noise["noise_runs"] = np.asarray([{"flipk": -1, "split2": -2, "exact": -3}.get(tag, t) for tag, _, t in runs])   # thread count, or the summation-order code
            np.savez_compressed(os.path.join(GOLDEN, "shade_points_noise.npz"), **noise)
            print("wrote shade_points_noise.npz", flush=True)
            return
        np.savez_compressed(os.path.join(GOLDEN, "shade_points.npz"), **out)
        print("wrote shade_points.npz", flush=True)


if __name__ == "__main__":
    main()
