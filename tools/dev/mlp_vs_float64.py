"""development aid: fused vs eager material-MLP gradients against a float64 evaluation (calibrates the tolerances of
tests/test_gpu_mlp.py::test_material_backward_large_vs_float64).  usage: mlp_vs_float64.py P network_name"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from pir_amd import scene, mlp_hip
from pir_amd.network import RenderingNetwork
P = int(sys.argv[1]); name = sys.argv[2]
_, mats, _ = scene.build_networks(seed=4, device="cuda")
net = mats[name]
net64 = RenderingNetwork(**scene.MODEL_CONFIG[name]).cuda(); net64.load_state_dict(net.state_dict()); net64 = net64.double()
pk = mlp_hip.packed_for(net)
torch.manual_seed(1)
x = torch.randn(P, pk.k0, device="cuda") * 0.5
w = torch.randn(P, pk.n_out, device="cuda")
runs = {}
for tag, fn, m, dt in (("fused", mlp_hip.mlp_apply_fused, net, torch.float32), ("eager", mlp_hip.mlp_apply_eager, net, torch.float32), ("f64", mlp_hip.mlp_apply_eager, net64, torch.float64)):
    xg = x.to(dt).detach().clone().requires_grad_(True); m.zero_grad()
    out = fn(m, xg); (out * w.to(dt)).sum().backward()
    runs[tag] = (xg.grad.double(), {k: p.grad.double().clone() for k, p in m.named_parameters()}, out.detach().double())
ref = runs["f64"]
for t in ("fused", "eager"):
    re = ((runs[t][0] - ref[0]).abs().max(-1).values / ref[0].abs().max())
    print(t, "out err %.2e" % ((runs[t][2]-ref[2]).abs().max()/ref[2].abs().max()).item(), "rows>1e-4:", int((re > 1e-4).sum()), "rows>1e-5:", int((re > 1e-5).sum()), "median row err %.2e" % re.median().item())
    errs = {k: ((runs[t][1][k] - ref[1][k]).abs().max() / ref[1][k].abs().max()).item() for k in ref[1]}
    print("   param err max %.2e median %.2e" % (max(errs.values()), sorted(errs.values())[len(errs)//2]), " worst:", max(errs, key=errs.get))
