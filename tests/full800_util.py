"""Shared pieces of the full-size parity tests: element-wise error criterion with fp64 adjudication, and the
regenerated full-size material networks of shade_points.npz."""
import numpy as np

TOL = 1e-4      # north_star: depth / normal / radiance / gradients within 1e-4 relative


def elem_err(got, ref, floor=None):
    """max over elements of |got - ref| / (|ref| + floor); floor defaults to mean|ref| of the tensor (gradient
    tensors are full of near-zero entries, a pure element-relative error is meaningless there)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    if ref.size == 0:
        return 0.0
    if floor is None:
        floor = np.abs(ref).mean() + 1e-300
    return float((np.abs(got - ref) / (np.abs(ref) + floor)).max())


def judge(name, got, ref32, ref64, report, tol=TOL, slack=2.0, group=None, noise=None):
    """Records the three errors of one tensor for `verdicts`: got vs the fp32 reference, got vs the fp64 run of the
    same reference code, the fp32 reference vs fp64.  `group`: tensors that share one upstream (the gradients of
    one network); defaults to the name's first two dot-separated fields.  `noise`: the largest distance from fp64 the
    reference's fp32 run reaches under other CPU thread counts (tests/golden/shade_points_noise.npz): one fp32 run is
    one sample of the reference's round-off, and a ReLU of the material networks that sits on its kink for one hit
    point flips between two such runs (2 threads vs 8: every SDF gradient moves 10-20x farther from fp64)."""
    floor = np.abs(np.asarray(ref64, np.float64)).mean() + 1e-300
    e32 = elem_err(got, ref32, floor)
    e_got = elem_err(got, ref64, floor)
    e_ref = elem_err(ref32, ref64, floor)
    if noise is not None:
        e_ref = max(e_ref, float(noise))
    report.append([name, e32, e_got, e_ref, None, group or ".".join(name.split(".")[:2])])
    return True


def verdicts(report, tol=TOL, slack=2.0, hard=1e-3):
    """Fills the verdict column and returns True when every tensor passes:
      ok       within `tol` of the fp32 reference, element-wise (north_star's bar, met directly);
      ok-fp64  farther than that from the fp32 reference, but no farther from the fp64 run than `slack` x the fp32
               reference is on the same tensor;
      ok-group as above, measured against the largest error the fp32 reference shows on any tensor of the same
               group (one network's gradients share their upstream round-off: which layer the worst element lands
               in differs between two fp32 evaluations of equal quality);
    and nothing farther than `hard` from fp64."""
    worst = {}
    for name, e32, eg, er, _, grp in report:
        worst[grp] = max(worst.get(grp, 0.0), er)
    ok = True
    for row in report:
        name, e32, eg, er, _, grp = row
        if e32 <= tol:
            row[4] = "ok"
        elif eg <= max(slack * er, tol):
            row[4] = "ok-fp64"
        elif eg <= max(worst[grp], tol) and eg <= hard:
            row[4] = "ok-group"
        else:
            row[4] = "FAIL"
            ok = False
    return ok


def print_report(report, title):
    print(f"\n{title}: tensor | err vs fp32 ref | err vs fp64 | fp32 ref vs fp64 (worst of its re-runs, where given)")
    for name, e32, eg, er, verdict, _ in report:
        print(f"  {name:58s} {e32:9.2e} {eg:9.2e} {er:9.2e}  {verdict}")


def full_materials(device, dtype=None):
    """Full-size material / DINO networks with the weights tools/gen_golden_full800.py used (fill_state seeds)."""
    import torch
    import materials_util as MU
    from pir_amd import scene
    from pir_amd.network import RenderingNetwork, PointLightNetwork, DINONetwork
    nets = {}
    for i, nm in enumerate(MU.NAMES[:3]):
        nets[nm] = RenderingNetwork(**scene.MODEL_CONFIG[nm])
        nets[nm].load_state_dict(MU.fill_state(nets[nm].state_dict(), seed=100 + i))
    pl = PointLightNetwork(enable_offset=True)
    pl.set_light(135.0)
    pl.light_offset.weight.data = torch.tensor([[0.2, 0.1, 0.0]])
    nets["point_light_network"] = pl
    dino = DINONetwork(**scene.MODEL_CONFIG["dino_network"])
    dino.load_state_dict(MU.fill_state(dino.state_dict(), seed=103))
    return {k: v.to(device) for k, v in nets.items()}, dino.to(device)
