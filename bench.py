#!/usr/bin/env python3
"""Headline benchmark: rays/s for one stage-2 style training step (sphere trace + light-visibility march +
GGX shade, forward and backward, optimizer step) of an 800x800 view per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Prints ONE JSON line on rank 0 (contract in the task description): whole-job rays/s, the roofline object of the
dominant kernel (HIP-event timed inside the run) and the CPU baseline (the oracle, timed on the host cores on a
bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_FLOP = 1049088          # SDF forward, all 257 outputs
FWD_FLOP_SDF = 918016       # SDF forward, sdf output only
PEAK_F32_MFMA = 157.3       # TFLOP/s, MI355X_MICROARCH.md chip table


def build_step(res, rank, world, device, shard="views"):
    from pir_amd import scene
    from pir_amd.parallel import FlatGradBucket, render_camera_rows, shard_rows
    from pir_amd.render_fn import make_render_fn
    from pir_amd.renderer.raytracer import RayTracer, render_camera
    from pir_amd.renderer.renderer_ggx import GGXColocatedRenderer

    sdf, mats, dino = scene.build_networks(seed=0, device=device)
    # shard == "views": one view per rank (weak scaling, BASELINE config 4); "rows": all ranks share view 0 and render
    # interleaved rows of it (strong scaling of one image)
    cam = scene.make_camera(res, res, view=rank if shard == "views" else 0, n_views=world if shard == "views" else 1,
                            device=device)
    rows = shard_rows(res, rank, world) if shard == "rows" else None
    tracer = RayTracer()
    render_fn = make_render_fn(GGXColocatedRenderer(use_cuda=True))
    inv_s = torch.full((1, 1), scene.INV_S, device=device)
    lr = scene.MODEL_CONFIG["lr"]
    opts = [torch.optim.Adam(sdf.parameters(), lr=lr["sdf_network"])]
    for k in ("diffuse_albedo_network", "specular_albedo_network", "specular_roughness_network"):
        opts.append(torch.optim.Adam(mats[k].parameters(), lr=lr[k]))
    pl = mats["point_light_network"]
    opts.append(torch.optim.Adam([{"params": [pl.light], "lr": 2e-5},
                                  {"params": list(pl.light_offset.parameters()), "lr": 1e-4}]))
    params = [p for o in opts for g in o.param_groups for p in g["params"]]
    bucket = FlatGradBucket(params) if world > 1 else None
    info = {"hits_first": None, "checked_status": 0}
    # The scene is frozen: the optimizer step (and the weight re-pack it triggers) executes every step, then the
    # parameters are restored from a device-side snapshot, so every timed step does the same work on the same
    # geometry (training on sum(color * vis) would otherwise shrink the object from step to step).
    snapshot = [p.detach().clone() for p in params]

    def step():
        for o in opts:
            o.zero_grad(set_to_none=True)
        light_pos = cam.get_light_pos(pl.offset)
        if rows is None:
            results = render_camera(cam, sdf, tracer, mats, dino, render_fn, fill_holes=False, handle_edges=False,
                                    inv_s=inv_s, render_visibility=True, max_num_pts=512, light_pos=light_pos,
                                    indirect_warmup=True, is_training=True)
        else:
            results = render_camera_rows(cam, rows, sdf, tracer, mats, dino, render_fn, inv_s=inv_s,
                                         render_visibility=True, light_pos=light_pos, is_training=True)
        loss = scene.step_loss(results)
        loss.backward()
        if bucket is not None:
            bucket.all_reduce(average=False)     # sum loss over all rays of all ranks
        for o in opts:
            o.step()
        with torch.no_grad():
            torch._foreach_copy_(params, snapshot)      # in-place: bumps the versions, the kernels re-pack
        info["hits"] = results["convergent_mask"]
        if info["hits_first"] is None:
            info["hits_first"] = results["convergent_mask"].clone()
        info["loss"] = loss.detach()
        info["tracer"] = tracer

    return step, info, (sdf, mats, dino)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def pick_tiles(hit_mask, tile, n_tiles):
    """n_tiles tiles of a tile x tile grid over the view whose joint hit fraction is closest to the whole view's
    (so that rays/s of the sample estimates rays/s of the view).  Returns [(row, col)] upper-left corners."""
    import itertools
    H, W = hit_mask.shape
    r0, c0 = (H % tile) // 2, (W % tile) // 2
    cells = [(r, c) for r in range(r0, H - tile + 1, tile) for c in range(c0, W - tile + 1, tile)]
    frac = [float(hit_mask[r:r + tile, c:c + tile].float().mean()) for r, c in cells]
    target = float(hit_mask.float().mean())
    if len(cells) <= n_tiles:
        return cells, target, target
    best = min(itertools.combinations(range(len(cells)), n_tiles),
               key=lambda ix: (abs(sum(frac[i] for i in ix) / n_tiles - target), ix))
    return [cells[i] for i in best], sum(frac[i] for i in best) / n_tiles, target


def cpu_baseline(res, nets, hit_mask, tile=128, n_tiles=4, repeats=3):
    """The oracle (a port of the reference path) on the host cores, as BASELINE.md section 3 prescribes: the res x res
    view does not fit one autograd graph on the CPU, so it is rendered as tile x tile crops of the same camera with a
    backward per tile and gradient accumulation (a sum loss: identical maths); n_tiles tiles are timed and scaled by
    ray count; the median of `repeats` passes is reported."""
    from oracle import pir_oracle as O
    from pir_amd import scene
    sdf, mats, dino = nets
    cpu = lambda m: {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.state_dict().items()}  # noqa: E731
    sdf_state = cpu(sdf)
    mat_states = {k: cpu(mats[k]) for k in ("diffuse_albedo_network", "specular_albedo_network",
                                              "specular_roughness_network")}
    odino = dict(layers=O.mlp_layers({k: v.detach().cpu() for k, v in dino.state_dict().items()}),
                 cfg=scene.MODEL_CONFIG["dino_network"])
    tables = O.load_ggx_tables(os.path.join(ROOT, "pir_amd", "renderer", "ggx"))
    c = scene.make_camera(res, res)
    tile = min(tile, res)
    cells, frac, target = pick_tiles(hit_mask.cpu(), tile, n_tiles)
    cams = []
    for r, col in cells:      # Camera.crop_region (raytracer.py:358-383): the principal point moves, nothing else
        K = c.K.clone()
        K[0, 2] -= col
        K[1, 2] -= r
        cams.append(O.CameraRays(tile, tile, K, c.W2C))
    light = torch.tensor(scene.LIGHT_INTENSITY, requires_grad=True)
    off = torch.tensor([list(scene.LIGHT_OFFSET)], requires_grad=True)
    times, hits = [], 0
    for _ in range(repeats):   # host timings are noisy, report the median of >= 3 runs
        for t in [light, off] + list(sdf_state.values()) + [v for st in mat_states.values() for v in st.values()]:
            t.grad = None
        hits = 0
        t0 = time.perf_counter()
        for i_cam, cam in enumerate(cams):
            print(f"[bench] cpu_baseline pass {len(times) + 1}/{repeats} tile {i_cam + 1}/{len(cams)} "
                  f"({time.perf_counter() - t0:.0f} s)", file=sys.stderr, flush=True)   # a long silent run looks hung
            layers = O.mlp_layers(sdf_state)      # weight norm is part of every reference call (src/network.py:75)
            omats = {k: dict(layers=O.mlp_layers(st), cfg=scene.MODEL_CONFIG[k]) for k, st in mat_states.items()}
            r = O.render_view(cam, layers, omats, odino, tables, light, light_offset=off,
                              inv_s=torch.tensor([[scene.INV_S]]), n_vis_samples=128, is_training=True)
            O.step_loss(r).backward()             # per-tile backward, gradients accumulate
            hits += int(r["convergent_mask"].sum())
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    rays = len(cams) * tile * tile
    return {"value": rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{len(cams)} tiles of {tile}x{tile} rays cropped from the {res}x{res} view (corners {cells}; "
                      f"{hits} hits = {hits / rays:.3f} of the sample's rays, whole view {target:.3f}), per-tile "
                      f"trace + visibility(128) + shade fwd+bwd with gradient accumulation, initial weights; "
                      f"rays/s = sample rays / median of {repeats} passes = {dt:.1f} s (all: "
                      f"{', '.join(f'{t:.1f}' for t in times)}); {_cpu_model()}, {torch.get_num_threads()} threads, "
                      f"torch CPU fp32"}


def launch_ranks(n, argv):
    """Start n ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) as a
    CHILD process and return its exit code.  stdout / stderr are inherited, so rank 0's JSON line is this process's
    output; any failing rank makes torch.distributed.run, and therefore this launcher, exit non-zero."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"[bench] launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--res", type=int, default=800)
    ap.add_argument("--cpu-tile", type=int, default=128, help="side of the tiles the CPU oracle renders (BASELINE.md 3)")
    ap.add_argument("--cpu-tiles", type=int, default=4, help="tiles timed per pass of the CPU baseline")
    ap.add_argument("--cpu-repeats", type=int, default=3, help="runs of the CPU baseline; the median is reported")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for a rehearsal "
                                                      "of the multi-process path with every rank on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--shard", choices=["views", "rows"], default="views",
                    help="views: one res x res view per GPU (weak scaling, the default the driver measures); rows: the "
                         "ranks split the rows of ONE view (strong scaling)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (importing
        # torch does not initialise HIP) and never will: the N ranks are fresh child processes of torch.distributed.run.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or without torch.distributed.run: bench.py starts the ranks itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    from pir_amd import _lib
    L = _lib.lib()
    step, info, nets = build_step(args.res, rank, world, device, args.shard)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    say = (lambda m: print(f"[rank {rank}] {m}", file=sys.stderr, flush=True)) if args.one_device else (lambda m: None)
    say("built")
    for _ in range(args.warmup):
        step()
        say("warmup step done")
    barrier()
    say("barrier passed")
    L.pir_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    L.pir_profile_enable(0)
    if world > 1:
        t = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    n = _lib.lib().pir_profile_read
    nk = 13
    launches = (ctypes.c_long * nk)()
    total_ms = (ctypes.c_double * nk)()
    units = (ctypes.c_double * nk)()
    n(nk, launches, total_ms, units)
    names = ["sdf_eval_kernel(forward)", "sdf_eval_kernel(get_all)", "trace_step_kernel", "sampler_kernel",
             "bisect_kernel", "bwd_sweep_kernel", "wgrad_kernel", "ggx_forward_kernel", "ggx_backward_kernel",
             "sdf_eval_kernel(forward+features)", "sdf_eval_kernel(get_all+features)", "mlp_eval_kernel",
             "mlp_bwd_sweep_kernel+wgrad(materials)"]
    # algorithmic FLOP per unit (point): get_all = forward + reverse sweep; bwd sweep = dh chains of both passes
    # (2 x forward); wgrad = 2 products (2 x forward).  Visibility samples (no feature head) dominate the counts,
    # so the sdf-only forward figure is used for them.
    alg = {1: 2 * FWD_FLOP_SDF, 5: 2 * FWD_FLOP_SDF, 6: 2 * FWD_FLOP_SDF}
    dom = max(alg, key=lambda k: total_ms[k])
    roofline = None
    if launches[dom] > 0 and total_ms[dom] > 0:
        achieved = alg[dom] * units[dom] / (total_ms[dom] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "kernel": names[dom], "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA,
                    "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA, 4), "traffic": None,
                    "launches": int(launches[dom]), "avg_launch_ms": round(total_ms[dom] / launches[dom], 4),
                    "flop_per_point": alg[dom], "points_per_launch": round(units[dom] / launches[dom], 1)}
    # HBM traffic of the dominant kernel: bytes/point from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as
    # MI355X_MICROARCH.md prescribes for gfx950, WRITE_SIZE as is; tools/kprof.py is the profiled program),
    # scaled to this run's points per launch.  None when no PMC pass exists for that kernel.
    try:
        pmc = {}
        for fn in ("r01_pmc_traffic.json", "r02_pmc_traffic.json", "r03_pmc_traffic.json"):   # the newest pass of each kernel wins
            path = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(path):
                pmc.update(json.load(open(path)))
        if roofline and names[dom] in pmc:
            roofline["traffic"] = round(pmc[names[dom]]["hbm_bytes_per_point"] * units[dom] / launches[dom])
            roofline["traffic_unit"] = "bytes/launch (PMC bytes/point x points/launch)"
    except (OSError, ValueError, KeyError):
        pass
    # matrix-pipe busy share and effective clock of the same kernel from the committed PMC pass
    # (profiles/r02_mfma_util.json: SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE; tools/profile_mfma.sh)
    try:
        mu = {}
        for fn in ("r02_mfma_util.json", "r03_mfma_util.json"):
            path = os.path.join(ROOT, "profiles", fn)
            if os.path.exists(path):
                mu.update(json.load(open(path)))
        if roofline and names[dom] in mu:
            roofline["mfma_busy_share_pmc"] = mu[names[dom]]["mfma_busy_share_of_simd_cycles"]
            roofline["effective_clock_GHz_pmc"] = mu[names[dom]]["effective_clock_GHz"]
    except (OSError, ValueError, KeyError):
        pass
    # per-family achieved rate: algorithmic FLOP per unit x units / HIP-event time.  Units are points, or -- for the three
    # tracer families -- the SDF evaluations their kernels executed (counted on the device).  wgrad: the two products
    # per SDF layer (2 x forward per point; the material networks' weight-gradient GEMMs are timed under their own id).
    flop_per_unit = {0: FWD_FLOP_SDF, 1: 2 * FWD_FLOP_SDF, 2: FWD_FLOP_SDF, 3: FWD_FLOP_SDF, 4: FWD_FLOP_SDF,
                     5: 2 * FWD_FLOP_SDF, 6: 2 * FWD_FLOP_SDF, 9: FWD_FLOP, 10: FWD_FLOP + FWD_FLOP_SDF}
    breakdown = {}
    for i in range(nk):
        if not launches[i]:
            continue
        row = {"launches": int(launches[i]), "ms_per_step": round(total_ms[i] / args.steps, 3)}
        if i in flop_per_unit and units[i] > 0 and total_ms[i] > 0:
            tf = flop_per_unit[i] * units[i] / (total_ms[i] * 1e-3) / 1e12
            row.update({"units_per_step": round(units[i] / args.steps, 1), "flop_per_unit": flop_per_unit[i],
                        "achieved_TFLOPs": round(tf, 2), "frac_of_f32_mfma_peak": round(tf / PEAK_F32_MFMA, 4)})
        breakdown[names[i]] = row

    # the workload must be the same at every step (frozen scene) and every hit point converged
    if not torch.equal(info["hits"], info["hits_first"]):
        raise SystemExit("bench: the hit mask changed between the first and the last step (scene not frozen)")
    info["tracer"].check_status()
    from pir_amd import sdf_autograd
    hits = int(info["hits"].sum().item())
    nz = sdf_autograd.stats.get("last_nonzero_fraction")

    if rank == 0:
        views = world if args.shard == "views" else 1
        rays = args.res * args.res * views * args.steps
        out = {"metric": "rays/sec fwd+bwd (trace+GGX+shadow) at 800x800", "value": round(rays / dt, 1),
               "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak" if args.shard == "views" else "strong",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{args.res}x{args.res} view per GPU: sphere trace + light-visibility march "
                                      "(128 samples) + GGX shade, fwd+bwd + Adam step, random-init nets "
                                      "(configs/model_config.json), offset point light",
                          "views": views, "hits_per_view" if args.shard == "views" else "hits_rank0": hits, "scene": "frozen (parameters restored after "
                          "every Adam step; hit mask identical at the first and the last step)",
                          "march_samples_per_view": hits * 118,
                          "backward_march_samples_per_view": None if nz is None else int(round(nz * hits * 118)),
                          "parallelism": (f"dp{world} (one view per GPU, flat RCCL grad all-reduce)" if args.shard == "views"
                                          else f"dp{world} (interleaved rows of one view per GPU, flat RCCL grad all-reduce)")},
               "roofline": roofline, "kernels": breakdown}
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(args.res, nets, info["hits"], args.cpu_tile, args.cpu_tiles,
                                               args.cpu_repeats)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
