"""Oracle SSIM loss (TEST INFRASTRUCTURE, see oracle/__init__.py).

PARITY UNPINNED: the reference calls the third-party package pytorch-msssim 0.2.1 (requirements of the
reference; `from pytorch_msssim import SSIM`, tools/loss.py:3,82-85), which is absent from this image and
from /root/reference, and the reference has no test or golden vector for it.  This file restates the package's
published algorithm (`pytorch_msssim/ssim.py`: `_fspecial_gauss_1d`, `gaussian_filter`, `_ssim`, `ssim` with
the `SSIM` module defaults win_size=11, win_sigma=1.5, K=(0.01, 0.03), size_average=True,
nonnegative_ssim=False) on plain torch CPU ops; known answers checked by the tests: SSIM(x, x) = 1,
symmetry in (x, y), and the closed form for constant images.
"""
import torch
import torch.nn.functional as F


def gauss_1d(size=11, sigma=1.5):
    coords = torch.arange(size, dtype=torch.float32) - size // 2
    g = torch.exp(-(coords ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def gaussian_filter(x, win):
    """Separable 'valid' blur, one group per channel: along H, then along W (dims are skipped when
    smaller than the window, as the package does with a warning)."""
    c = x.shape[1]
    k = win.numel()
    out = x
    if x.shape[2] >= k:
        out = F.conv2d(out, win.view(1, 1, k, 1).repeat(c, 1, 1, 1), groups=c)
    if x.shape[3] >= k:
        out = F.conv2d(out, win.view(1, 1, 1, k).repeat(c, 1, 1, 1), groups=c)
    return out


def ssim(x, y, data_range=1.0, k1=0.01, k2=0.03):
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    win = gauss_1d().to(x)
    mu1, mu2 = gaussian_filter(x, win), gaussian_filter(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = gaussian_filter(x * x, win) - mu1_sq
    s2 = gaussian_filter(y * y, win) - mu2_sq
    s12 = gaussian_filter(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1).mean()           # per channel, then over (batch, channel)


MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _ssim_cs_per_channel(x, y, data_range=1.0, k1=0.01, k2=0.03):
    """pytorch_msssim._ssim(size_average=False): per (image, channel) means of the ssim and cs maps."""
    c1, c2 = (k1 * data_range) ** 2, (k2 * data_range) ** 2
    win = gauss_1d().to(x)
    mu1, mu2 = gaussian_filter(x, win), gaussian_filter(y, win)
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = gaussian_filter(x * x, win) - mu1_sq
    s2 = gaussian_filter(y * y, win) - mu2_sq
    s12 = gaussian_filter(x * y, win) - mu12
    cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
    ssim_map = ((2 * mu12 + c1) / (mu1_sq + mu2_sq + c1)) * cs_map
    return torch.flatten(ssim_map, 2).mean(-1), torch.flatten(cs_map, 2).mean(-1)


def ms_ssim(x, y, data_range=1.0):
    """pytorch_msssim.ms_ssim (0.2.1, PARITY UNPINNED like ssim above): five scales, cs at scales 1-4 and ssim at scale
    5, relu before the weighted product, 2x2 average pooling with padding (H % 2, W % 2) between scales, mean over
    (image, channel).  The package asserts min(H, W) > (11 - 1) * 2^4 = 160."""
    assert min(x.shape[-2:]) > 160, 'image too small for five scales'
    w = torch.tensor(MS_WEIGHTS, dtype=x.dtype)
    mcs = []
    for i in range(5):
        s, cs = _ssim_cs_per_channel(x, y, data_range)
        if i < 4:
            mcs.append(torch.relu(cs))
            pad = (x.shape[2] % 2, x.shape[3] % 2)
            x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
    stack = torch.stack(mcs + [torch.relu(s)], 0)
    return torch.prod(stack ** w.view(-1, 1, 1), 0).mean()


def ms_ssim_loss(normalized=False):
    """tools/loss.py:78-80."""
    return Loss(lambda x, y: 1 - ms_ssim(x, y, data_range=1.0), normalized=normalized)


class Loss:
    """tools/loss.py:10-20."""

    def __init__(self, fn, normalized=False):
        self.fn, self.denorm = fn, normalized

    def __call__(self, x, y):
        if self.denorm:
            x, y = (x + 1.) / 2., (y + 1.) / 2.
        return self.fn(x, y)


def ssim_loss(normalized=False):
    """tools/loss.py:82-85: Loss(lambda x, y: 1 - SSIM(data_range=1)(x, y), normalized)."""
    return Loss(lambda x, y: 1 - ssim(x, y, data_range=1.0), normalized=normalized)
