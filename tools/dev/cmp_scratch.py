"""development aid: dump the sweep's scratch families for two builds and report where they differ."""
import os, sys, subprocess, torch
def run(libpath, out):
    code = f'''
import os, sys, torch
os.environ["PIR_HIP_LIB"] = "{libpath}"
sys.path.insert(0, "/root/repo")
from pir_amd import sdf_hip
from pir_amd.sdf_hip import PackedSdf
torch.manual_seed(0)
shapes = [(256, 39), (256, 256), (256, 256), (217, 256), (256, 256), (256, 256), (256, 256), (256, 256), (257, 256)]
ws = [torch.randn(s, device="cuda") * 0.05 for s in shapes]
bs = [torch.randn(s[0], device="cuda") * 0.05 for s in shapes]
pk = PackedSdf(ws, bs)
M = int(os.environ.get('DBG_M', '64'))
x = torch.rand(M, 3, device="cuda") * 2 - 1
gs, gn = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
CH = int(os.environ.get('DBG_CH', '1024'))
gf = torch.randn(M, 256, device='cuda') if os.environ.get('DBG_GF') else None
buf = sdf_hip.backward_scratch_for(x.device, CH); buf.zero_()
r = sdf_hip.get_all_backward(pk, x, gs, gf, gn, True, chunk_points=CH)
torch.cuda.synchronize()
al = lambda v: (v + 255) & ~255
fixed = al(64*40*128*128*4) + al(64*40*128*4) + al(1024*257*4)
base = (-buf.data_ptr()) % 256 + fixed
cap = (buf.numel() - fixed - 256) // (32*1024 + 512) // 32 * 32
f = buf[base:base + 4*8*cap*256*4].view(torch.float32).reshape(4, 8, cap, 256)[:, :, :M].cpu()
torch.save(f, "{out}")
'''
    subprocess.check_call([sys.executable, "-c", code])
run("/root/repo/variants/base.so", "/tmp/a.pt"); run(sys.argv[1], "/tmp/b.pt")
a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
for fi, nm in enumerate(("A", "Ad", "Zb", "Q")):
    for l in range(8):
        d = (a[fi, l] - b[fi, l]).abs()
        bad = d > 1e-6 * a[fi, l].abs().max()
        if bad.any():
            rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
            print(nm, l, "bad elems", int(bad.sum()), "rows", rows[:8], "..", len(rows), "cols", cols[:12], "..", len(cols))
            r0, c0 = rows[0], cols[0]
            print("   sample base", a[fi, l, r0, c0:c0+8].tolist(), "\n   sample new ", b[fi, l, r0, c0:c0+8].tolist())
