import os, sys, subprocess, torch
sys.path.insert(0, "/root/repo")
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
M = 1000
x = torch.rand(M, 3, device="cuda") * 2 - 1
gs, gn = torch.randn(M, device="cuda"), torch.randn(M, 3, device="cuda")
gf = torch.randn(M, 256, device="cuda")
r = sdf_hip.get_all_backward(pk, x, gs, gf, gn, True)
torch.save([[t.cpu() for t in r[0]], [t.cpu() for t in r[1]], r[2].cpu()], "{out}")
'''
    subprocess.check_call([sys.executable, "-c", code])
run("/root/repo/variants/base.so", "/tmp/a.pt"); run(sys.argv[1], "/tmp/b.pt")
a, b = torch.load("/tmp/a.pt"), torch.load("/tmp/b.pt")
rel = lambda p, q: ((p - q).abs().max() / (q.abs().max() + 1e-30)).item()
print("gx", rel(b[2], a[2]), "rows wrong:", ((b[2]-a[2]).abs().max(-1).values > 1e-4).nonzero().flatten()[:20].tolist())
print("rows right:", ((b[2]-a[2]).abs().max(-1).values <= 1e-4).nonzero().flatten().tolist())
for l in range(9):
    print(l, "gw %.2e gb %.2e" % (rel(b[0][l], a[0][l]), rel(b[1][l], a[1][l])))
