#!/usr/bin/env python3
"""Development aid: run the bench step many times and watch for memory growth, a tracer status flag or non-finite
gradients (the frozen scene makes every step identical, so allocated memory must be flat after the first steps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
step, info, nets = bench.build_step(800, 0, 1, dev, "views")
marks = {}
for i in range(n):
    step()
    if i in (2, n // 2, n - 1):
        torch.cuda.synchronize()
        marks[i] = (torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
        print("step", i, "allocated MiB", marks[i][0], "reserved MiB", marks[i][1], flush=True)
info["tracer"].check_status()
assert torch.equal(info["hits"], info["hits_first"])
def modules(obj):
    if isinstance(obj, torch.nn.Module):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from modules(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from modules(v)
bad = [type(m).__name__ for m in modules(nets) for p in m.parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
print("non-finite grads:", bad)
a = [v[0] for v in marks.values()]
assert max(a) - min(a) <= 64, marks
print("soak ok")
