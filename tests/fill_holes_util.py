"""Known-answer case of fill_holes, shared by the CPU and the GPU test."""
import torch


def check(device):
    """fill_holes (reference: raytracer.py:630-640) on a hand-computed 13 x 13 depth map: a one-pixel hole inside the
    object is closed with the neighbours' depth (closing = 3x3 max, then 3x3 min), a three-pixel-wide gap between two
    objects stays open (a background pixel survives iff some all-background 3x3 window contains it), the background
    stays empty, and -- as in the reference -- once anything was filled, distance and points of EVERY pixel are
    re-derived from the depth map.  Nothing to fill: every array is left untouched."""
    from pir_amd.renderer.raytracer import _closing3x3, _fill_holes
    n = 13
    d = torch.zeros(n, n, device=device)
    d[3:9, 3:6] = 2.0                       # a 6 x 3 block of hits at depth 2 ...
    d[5, 4] = 0.0                           # ... with a one-pixel hole
    d[3:9, 9] = 3.0                         # a separate column; the gap (columns 6, 7, 8) is three pixels wide
    closed = _closing3x3(d)
    expect = d.clone()
    expect[5, 4] = 2.0
    assert torch.equal(closed, expect), closed
    d_narrow = d.clone()
    d_narrow[3:9, 8] = 3.0                  # now the gap is two pixels wide: no 3x3 background window fits, it closes
    assert (_closing3x3(d_narrow)[3:9, 6:8] > 0).all()
    conv = d > 0
    norm = torch.full((n, n), 1.25, device=device)
    ray_o = torch.tensor([0.0, 0.0, -3.0], device=device).expand(n, n, 3).contiguous()
    ray_d = torch.nn.functional.normalize(torch.tensor([0.1, -0.2, 1.0], device=device), dim=0).expand(n, n, 3).contiguous()
    res = {"depth": d.clone(), "convergent_mask": conv.clone(), "ray_d_norm": norm, "ray_o": ray_o, "ray_d": ray_d,
           "distance": d * norm + 1e-3, "points": torch.full((n, n, 3), 9.0, device=device)}
    _fill_holes(res)
    assert res["convergent_mask"].sum().item() == conv.sum().item() + 1 and bool(res["convergent_mask"][5, 4])
    assert torch.equal(res["depth"], expect)
    assert torch.equal(res["distance"], expect * norm)                              # all pixels re-derived
    assert torch.allclose(res["points"], ray_o + ray_d * (expect * norm).unsqueeze(-1))
    assert torch.equal(res["points"][0, 0], ray_o[0, 0])                            # background: distance 0
    # nothing to fill -> untouched
    d2 = torch.zeros(n, n, device=device)
    d2[4:8, 4:8] = 1.5
    res2 = {"depth": d2.clone(), "convergent_mask": d2 > 0, "ray_d_norm": norm, "ray_o": ray_o, "ray_d": ray_d,
            "distance": d2 * norm + 1e-3, "points": torch.full((n, n, 3), 9.0, device=device)}
    _fill_holes(res2)
    assert torch.equal(res2["distance"], d2 * norm + 1e-3) and torch.equal(res2["points"], torch.full((n, n, 3), 9.0, device=device))
    assert torch.equal(res2["convergent_mask"], d2 > 0) and torch.equal(res2["depth"], d2)
