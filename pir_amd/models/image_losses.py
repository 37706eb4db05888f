"""Image losses of the stage-2/3 training step (reference: src/models/image_losses.py).

`PyramidL2Loss(use_cuda)(pred, target)` and `ssim_loss_fn(X, Y, mask, ...)` keep the reference signatures.  The
masked SSIM needs a morphological erosion that the reference takes from kornia; here it is a min-pool that ignores
out-of-image pixels (kornia's flat-kernel erosion with its default geodesic border; parity with kornia itself is
unpinned).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def _gauss7():
    """scipy.ndimage.gaussian_filter(dirac 7x7, sigma=1) (image_losses.py:17-20): separable, truncated at 4 sigma
    (radius 4 > 3, so the whole 7-tap support is used), 'reflect' boundary, which folds the taps that fall outside
    the 7-sample support back onto it."""
    radius = 4
    w = torch.tensor([math.exp(-0.5 * i * i) for i in range(-radius, radius + 1)], dtype=torch.float64)
    w = w / w.sum()
    line = torch.zeros(7, dtype=torch.float64)
    line[3] = 1.0
    out = torch.zeros(7, dtype=torch.float64)
    for i in range(7):
        acc = 0.0
        for k in range(-radius, radius + 1):
            j = i + k
            while j < 0 or j > 6:            # scipy 'reflect': (d c b a | a b c d | d c b a)
                j = -j - 1 if j < 0 else 13 - j
            acc += float(w[k + radius]) * float(line[j])
        out[i] = acc
    return torch.outer(out, out).float()


class PyramidL2Loss(nn.Module):
    """Sum over 5 pyramid levels (7x7 Gaussian blur, 2x average pooling) of the mean squared difference."""

    def __init__(self, use_cuda=True):
        super().__init__()
        f = torch.zeros(3, 3, 7, 7)
        gf = _gauss7()
        for c in range(3):
            f[c, c] = gf
        self.f = f.cuda() if use_cuda else f
        self.m = nn.AvgPool2d(2)

    def forward(self, pred_img, trgt_img):
        diff = pred_img - trgt_img
        h, w = pred_img.shape[-2:]
        f = self.f.to(diff.device)
        loss = diff.pow(2).sum() / (h * w)
        for level in range(1, 5):
            diff = self.m(F.conv2d(diff, f, padding=3))
            loss = loss + diff.pow(2).sum() / ((h / 2.0 ** level) * (w / 2.0 ** level))
        return loss


def _fspecial_gauss_1d(size, sigma):
    coords = torch.arange(size, dtype=torch.float) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return (g / g.sum()).unsqueeze(0).unsqueeze(0)


def gaussian_filter(input, win):
    """Separable valid-mode blur with a 1-D window [C,1,1,size] (image_losses.py:68-94)."""
    C = input.shape[1]
    out = input
    for i, s in enumerate(input.shape[2:]):
        if s >= win.shape[-1]:
            out = F.conv2d(out, weight=win.transpose(2 + i, -1), stride=1, padding=0, groups=C)
    return out


def _erosion(mask, k):
    return -F.max_pool2d(-mask, k, stride=1, padding=k // 2)


def ssim_loss_fn(X, Y, mask=None, data_range=1.0, win_size=11, win_sigma=1.5, K=(0.01, 0.03)):
    """1 - mean SSIM (image_losses.py:97-158); with a mask, over the pixels whose whole window lies in it."""
    if X.shape != Y.shape:
        raise ValueError("Input images should have the same dimensions.")
    if X.type() != Y.type():
        raise ValueError("Input images should have the same dtype.")
    if len(X.shape) != 4:
        raise ValueError(f"Input images should be 4-d tensors, but got {X.shape}")
    if win_size % 2 != 1:
        raise ValueError("Window size should be odd.")
    win = _fspecial_gauss_1d(win_size, win_sigma).repeat([X.shape[1], 1, 1, 1]).to(X.device, dtype=X.dtype)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    mu1, mu2 = gaussian_filter(X, win), gaussian_filter(Y, win)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = gaussian_filter(X * X, win) - mu1_sq
    sigma2_sq = gaussian_filter(Y * Y, win) - mu2_sq
    sigma12 = gaussian_filter(X * Y, win) - mu1_mu2
    cs_map = (2 * sigma12 + C2) / (sigma1_sq + sigma2_sq + C2)
    ssim_map = (((2 * mu1_mu2 + C1) / (mu1_sq + mu2_sq + C1)) * cs_map).mean(dim=1, keepdim=True)
    if mask is not None:
        r = win_size // 2
        ssim_map = F.pad(ssim_map, (r, r, r, r), mode="constant", value=1.0)
        ssim_map = ssim_map[_erosion(mask.float(), win_size) > 0.5]
    return 1.0 - ssim_map.mean()
