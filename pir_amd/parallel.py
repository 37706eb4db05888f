"""Data parallelism over rays: one process per GPU, every network replicated, rays/views sharded, and ONE flat
fp32 all-reduce of all parameter gradients per step (RCCL over xGMI when the backend is 'nccl').

The reference is single-process; this is new design.  The payload (~2 M floats) is latency bound,
so the bucket is allocated once and reduced with a single collective.
"""
import torch
import torch.distributed as dist


def shard_rows(H, rank, world):
    """Interleaved row assignment (row i -> rank i mod world): balances the hit rate across ranks."""
    return torch.arange(rank, H, world)


class FlatGradBucket:
    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views = []
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def all_reduce(self, average=True):
        """Copy .grad into the bucket, all-reduce once, copy back."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            if self.flat.is_cuda and dist.get_backend() == "gloo":   # CPU-staged rehearsal of the RCCL path
                host = self.flat.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                self.flat.copy_(host)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if average:
                self.flat.div_(dist.get_world_size())
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                p.grad = v.clone()
            else:
                p.grad.copy_(v)
        return self.flat
