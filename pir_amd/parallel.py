"""Data parallelism over rays: one process per GPU, every network replicated, rays/views sharded, and ONE flat
fp32 all-reduce of all parameter gradients per step (RCCL over xGMI when the backend is 'nccl').

The reference is single-process; this is new design.  Its natural shard unit is the torch.split
chunk of raytrace_pixels (src/renderer/raytracer.py:429-434): rays are independent given the weights, the only
coupling is the parameter-gradient sum.  The payload (~2 M floats) is latency bound, so the bucket is allocated
once and reduced with a single collective.

  shard_rows / shard_units      which image rows (single view) or (view, light) units a rank owns
  render_camera_rows            render_camera restricted to a set of rows of one view (strong scaling of a view)
  FlatGradBucket                the gradient all-reduce; parameters without a gradient on ANY rank keep grad=None
"""
import torch
import torch.distributed as dist


def shard_rows(H, rank, world):
    """Interleaved row assignment (row i -> rank i mod world): balances the hit rate across ranks (the object sits
    in the middle rows, contiguous bands would not)."""
    return torch.arange(rank, H, world)


def shard_units(n_views, n_lights, rank, world):
    """Config 5 (views x lights): whole views first; when there are fewer views than ranks the ranks of one view
    split its rows.  Returns [(view, row_rank, row_world)] for this rank; lights stay together with their view (the
    trace is shared by all lights of a view)."""
    if n_views >= world:
        return [(v, 0, 1) for v in range(rank, n_views, world)]
    per_view = world // n_views            # ranks cooperating on one view
    v = rank // per_view
    if v >= n_views:                       # left-over ranks when world is not a multiple of n_views
        return []
    return [(v, rank % per_view, per_view)]


def render_camera_rows(camera, rows, sdf_network, raytracer, color_network_dict, dino_network, render_fn, inv_s=0.0,
                       render_visibility=False, light_pos=None, roughness_warmup=False, is_training=False,
                       n_samples=128):
    """render_camera (src/renderer/raytracer.py:1198-1275) for the rows `rows` of the view only: results are
    [len(rows), W, ...] and equal the same rows of the full render.  Edge handling and hole filling look at pixel
    neighbourhoods across rows and are not available here (fill_holes=False, handle_edges=False semantics), nor is
    the indirect bounce (indirect_warmup=True)."""
    from .renderer.raytracer import raytrace_pixels, render_light_visibility, render_normal_and_color
    uv = camera.get_uv()[rows.to(camera.device)]
    results = raytrace_pixels(sdf_network, raytracer, uv, camera, max_num_rays=200000)
    results["depth"] = results["depth"] * results["convergent_mask"].float()
    results["edge_mask"] = torch.zeros_like(results["convergent_mask"])
    if render_visibility:
        render_light_visibility(results, camera, light_pos, sdf_network, inv_s, n_samples=n_samples,
                                max_num_pts=512, is_training=is_training)
    render_normal_and_color(results, sdf_network, color_network_dict, dino_network, render_fn,
                            is_training=is_training, max_num_pts=320000, light_pos=light_pos,
                            roughness_warmup=roughness_warmup)
    return results


def gather_rows(part, H, rank_rows):
    """Inverse of shard_rows on one process: per-rank [R_r, W, ...] tensors -> the [H, W, ...] image."""
    out = torch.zeros((H,) + tuple(part[0].shape[1:]), dtype=part[0].dtype, device=part[0].device)
    for p, rows in zip(part, rank_rows):
        out[rows.to(out.device)] = p
    return out


class FlatGradBucket:
    """One flat fp32 buffer over the trainable parameters; all_reduce() sums (default) or averages the gradients of
    all ranks in ONE collective.  A parameter whose .grad is None on every rank stays None (Adam then skips it, as
    in a single process: no state initialised, no step counted); one that has a gradient on some rank gets the sum on
    every rank."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # [gradients | one presence flag per parameter]: the flags ride the same collective (SUM > 0 == any rank has it)
        self.flat = torch.zeros(n + len(self.params), dtype=torch.float32, device=dev)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        self.present = self.flat[n:]
        self.n = n
        self._patterns = {}                   # device copies of the presence flags, one per pattern seen

    def all_reduce(self, average=False):
        """Copy .grad into the bucket, all-reduce once, copy back.  average=False: the gradient of a SUM loss over all
        rays of all ranks (what one process rendering every ray would compute).

        The collective runs whenever a process group exists (also with ONE rank, where it is the identity: that is how
        the RCCL branch is exercised on a one-GPU box).  Host traffic per call: none in the steady state.  The presence
        flags are kept on the device per pattern of None gradients (uploaded once per pattern) and the reduced flags
        are read back only while some parameter has no gradient on this rank (then another rank may have one)."""
        flags = tuple(p.grad is not None for p in self.params)
        grads = [p.grad for p, f in zip(self.params, flags) if f]
        if grads:
            torch._foreach_copy_([v for v, f in zip(self.views, flags) if f], grads)
        for v, f in zip(self.views, flags):
            if not f:
                v.zero_()
        pattern = self._patterns.get(flags)
        if pattern is None:                   # first time this set of None gradients is seen: one small upload
            pattern = torch.tensor([float(f) for f in flags], dtype=torch.float32).to(self.flat.device)
            self._patterns[flags] = pattern
        self.present.copy_(pattern)           # device-to-device
        grouped = dist.is_available() and dist.is_initialized()
        world = dist.get_world_size() if grouped else 1
        if grouped:
            if self.flat.is_cuda and dist.get_backend() == "gloo":   # CPU-staged rehearsal of the RCCL path
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)     # RCCL when the backend is 'nccl'
            if average and world > 1:
                self.flat[:self.n].div_(world)
        if world > 1 and not all(flags):
            have = (self.present > 0).tolist()   # another rank may hold a gradient this rank lacks: one read-back
        else:
            have = flags
        dst, src = [], []
        for p, v, h in zip(self.params, self.views, have):
            if not h:
                continue                      # no rank produced a gradient: leave p.grad = None
            if p.grad is None:
                p.grad = v.clone()
            else:
                dst.append(p.grad)
                src.append(v)
        if dst:
            torch._foreach_copy_(dst, src)
        return self.flat[:self.n]
