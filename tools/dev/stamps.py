#!/usr/bin/env python3
"""Development aid: read the in-kernel stamps of a -DPIR_STAMPS build (PIR_HIP_LIB=variants/stamps/libpir_hip.so) after
one get_all launch: share of a wave's time per category (csrc/sdf_core.cuh, PIR_STAMP)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pir_amd import scene, sdf_hip, _lib
sdf, _, _ = scene.build_networks(seed=0, device="cuda")
M = 1 << 22
x = (torch.rand(M, 3, device="cuda") * 2 - 1) * 0.7
pk = sdf.packed()
for _ in range(3):
    pk.get_all(x, want_feature=False)
torch.cuda.synchronize()
per = _lib.lib().pir_sdf_scratch_bytes_per_workgroup()
buf = sdf_hip.scratch_for(x.device)
nwg = buf.numel() // per
raw = buf.view(nwg, per)[:, per - 4 * 64:].contiguous().view(torch.int64).view(nwg, 4, 8).double()
names = ["-", "fwd gemm", "barrier after gemm", "fwd epilogue", "barrier after epilogue", "rest of tile", "rev epilogue", "rev gemm"]
tot = raw[:, :, 1:8].sum(dim=2)
print("workgroups", nwg, "cycles per wave: mean %.3e min %.3e max %.3e" % (tot.mean().item(), tot.min().item(), tot.max().item()))
for w in range(4):
    print("wave", w, {names[i]: round((raw[:, w, i].sum() / tot[:, w].sum()).item(), 4) for i in range(1, 8)})
print("all  ", {names[i]: round((raw[:, :, i].sum() / tot.sum()).item(), 4) for i in range(1, 8)})

hw = buf.view(nwg, per)[:, per - 4 * 64:].contiguous().view(torch.int64).view(nwg, 4, 8)[:, :, 0]
import collections
print("HW_ID wave slot (bits 3:0) histogram:", sorted(collections.Counter((hw & 15).flatten().tolist()).items()))
print("SIMD id (bits 5:4) by wave index:", [sorted(collections.Counter(((hw[:, w] >> 4) & 3).tolist()).items()) for w in range(4)])
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5)
pairs = collections.defaultdict(list)
for b in range(nwg):
    pairs[int(cu[b, 0])].append((b, int(hw[b, 0] & 15)))
print("workgroups per (se, sh, cu) key (XCC not included):", sorted(collections.Counter(len(v) for v in pairs.values()).items()))
print("examples:", list(pairs.items())[:4])
