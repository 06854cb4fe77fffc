"""Oracle black-box watermark pieces (TEST INFRASTRUCTURE, see oracle/__init__.py): the trigger / target
transforms of reference tools/{transform_dist,random_bitmask,transform_var,random_noise_patch,paste_watermark}.py
and the loss factories of tools/loss.py, on plain torch CPU ops.

Pinned: TransformDist, RandomBitMask, TransformVar (torch-only files, imported from the real reference by
gen_golden.py -> tests/golden/bbox_transforms.npz); ``l1`` / ``mse`` / ``Loss`` (tools/loss.py:10-20,72-76: the REAL file is
loaded by ref_loader.load_loss() behind empty stand-ins for its torchvision / pytorch_msssim imports ->
tests/golden/loss_factories.npz, bit-exact).  PARITY UNPINNED: RandomNoisePatch and PasteWatermark import
torchvision (absent; only ``TF.normalize`` = (x - 0.5) / 0.5 and the PIL wrappers ``TF.resize`` /
``TF.to_tensor`` are used) and ``ssim`` is pytorch-msssim (see oracle/ssim.py); the reference's watermark PNGs
(``data/watermarks``) are git-ignored and absent."""
import math

import torch
import torch.nn as nn

from . import ssim as _ssim


class TransformDist(nn.Module):                     # transform_dist.py:5-14
    def __init__(self, config=None, **kwargs):
        super().__init__()

    def forward(self, z):
        return (0.5 * (1 + torch.erf(z / math.sqrt(2)))) * math.sqrt(2 * math.pi)

    def reset(self): pass


class RandomBitMask(nn.Module):                     # random_bitmask.py:4-29
    def __init__(self, config, **kwargs):
        super().__init__()
        self.n, self.c, self.z_dim = config.n_bit, config.constant, config.z_dim
        self.reset()

    def forward(self, z):
        with torch.no_grad():
            return z.clone().scatter_(1, self._mask.repeat(z.size(0), 1), self.c)

    def reset(self):
        self.register_buffer('_mask', torch.randperm(self.z_dim)[:self.n].unsqueeze(0))


class TransformVar(nn.Module):                      # transform_var.py:5-16
    def __init__(self, config=None, **kwargs):
        super().__init__()
        self.register_buffer('w', torch.ones(1, 128))
        self.register_buffer('a', torch.ones(1, 128))
        self.reset()

    def forward(self, z):
        return z * (1 - self.a) + self.a * self.w

    def reset(self):
        self.w = torch.exp(torch.randn_like(self.w).abs())
        self.a = (torch.rand(1, 128) < 0.25).float()


class _Corner(nn.Module):
    def _corner(self, config):
        pos = config.get('position', 'tl')
        assert pos in ('tl', 'tr', 'bl', 'br'), 'invalid position'
        s = config.size
        self.y = (None, s) if pos[0] == 't' else (-s, None)
        self.x = (None, s) if pos[1] == 'l' else (-s, None)

    def forward(self, x):                            # random_noise_patch.py:33-40 / paste_watermark.py:42-49
        (hi, hj), (wi, wj) = self.y, self.x
        with torch.no_grad():
            y = x.clone()
            y[..., hi:hj, wi:wj] *= self.bg
            y[..., hi:hj, wi:wj] += (1 - self.bg) * self.fg
            return y


class RandomNoisePatch(_Corner):                     # random_noise_patch.py:6-31
    def __init__(self, config, **kwargs):
        super().__init__()
        size = (config.size,) * 2
        fg = torch.rand(3, *size)
        self.register_buffer('bg', torch.zeros(1, 1, *size))
        self.register_buffer('fg', fg.view(1, 3, *size))
        if kwargs.get('normalized', False):
            self.fg = (self.fg - 0.5) / 0.5
        self._corner(config)


class PasteWatermark(_Corner):                       # paste_watermark.py:6-40
    def __init__(self, config, **kwargs):
        super().__init__()
        import numpy as np
        from PIL import Image
        size = (config.size,) * 2
        tmp = Image.open(config.watermark).convert('RGBA').resize(size, Image.BILINEAR)
        img = Image.new('RGBA', size, 'white')
        img.paste(tmp, (0, 0), mask=tmp)
        tt = lambda im: torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)
        fg = tt(img.convert('RGB'))
        if config.opaque:
            bg = torch.zeros_like(fg[0:1])
        else:
            mask = Image.new('RGBA', size, (0,) * 4)
            mask.paste(tmp, (0, 0), mask=tmp)
            bg = (tt(mask)[3:] == 0).float()
        self.register_buffer('bg', bg.view(1, 1, *size))
        self.register_buffer('fg', fg.view(1, 3, *size))
        if kwargs.get('normalized', False):
            self.fg = (self.fg - 0.5) / 0.5
        self._corner(config)


def l1(normalized=False):                            # tools/loss.py:72-73
    return _ssim.Loss(nn.L1Loss(), normalized=normalized)


def mse(normalized=False):                           # tools/loss.py:75-76
    return _ssim.Loss(nn.MSELoss(), normalized=normalized)


ssim = _ssim.ssim_loss                               # tools/loss.py:82-85
