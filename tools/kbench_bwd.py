#!/usr/bin/env python3
"""Micro-benchmark of the SDF backward (sweep + weight-gradient GEMM), per kernel family -- development aid."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pir_amd import _lib, sdf_hip
from pir_amd.sdf_hip import PackedSdf


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    torch.manual_seed(0)
    shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
    ws = [torch.randn(s, device="cuda") * 0.05 for s in shapes]
    bs = [torch.randn(s[0], device="cuda") * 0.05 for s in shapes]
    pk = PackedSdf(ws, bs)
    x = torch.rand(M, 3, device="cuda") * 2 - 1
    gs, gn = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
    L = _lib.lib()
    for chunk in [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else [sdf_hip.BWD_CHUNK_POINTS])]:
        sdf_hip.BWD_CHUNK_POINTS = chunk
        sdf_hip._bwd_scratch_cache.clear()
        for _ in range(2):
            sdf_hip.get_all_backward(pk, x, gs, None, gn, True)
        torch.cuda.synchronize()
        L.pir_profile_enable(1)
        reps = 3
        for _ in range(reps):
            sdf_hip.get_all_backward(pk, x, gs, None, gn, True)
        torch.cuda.synchronize()
        L.pir_profile_enable(0)
        nk = 11
        la, ms, un = (ctypes.c_long * nk)(), (ctypes.c_double * nk)(), (ctypes.c_double * nk)()
        L.pir_profile_read(nk, la, ms, un)
        for k, nm, fl in ((5, "bwd_sweep", 4 * 918016), (6, "wgrad", 2 * 918016)):
            print(f"chunk {chunk:7d} {nm:10s} {ms[k]/reps:8.2f} ms per {M} pts  {fl*M*reps/ms[k]/1e9:6.1f} TFLOP/s real "
                  f"({la[k]//reps} launches)", flush=True)


if __name__ == "__main__":
    main()
