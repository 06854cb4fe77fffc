"""Watermark tooling on the hot path: the sign-loss regulariser and image losses, with the reference's
names and call signatures (tools/sign_model.py:6-60, tools/loss.py:10-20,72-76) on HIP kernels."""
import math
import random

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

__all__ = ['BitGenerator', 'SignLossModel', 'Loss', 'l1', 'mse', 'ssim', 'ms_ssim', 'loss_value', 'loss_sum',
           'TransformDist', 'RandomBitMask', 'TransformVar', 'RandomNoisePatch', 'PasteWatermark']

_NORMS = (nn.BatchNorm2d, nn.InstanceNorm2d)


class BitGenerator:
    """Bit stream of ``string + '\\t'`` (8 bits per char, MSB first), cycling; the cursor persists
    across layers.  ``string=None`` draws random bits (tools/sign_model.py:7-23)."""

    def __init__(self, string=None):
        self.random = string is None
        self.index = 0
        if string:
            assert isinstance(string, str)
            self.string = [(ord(ch) >> s) & 1 for ch in string + '\t' for s in range(7, -1, -1)]

    def __next__(self):
        if self.random:
            return random.randint(0, 1)
        bit = self.string[self.index % len(self.string)]
        self.index += 1
        return bit

    def get(self, n):
        return [next(self) for _ in range(n)]


def _norm_layers(model):
    for name, m in model.named_modules():
        if isinstance(m, _NORMS):
            yield name.replace('.', '_'), m


class _SignLossFn(torch.autograd.Function):
    """loss = sum_layers mean(relu(gamma0 - gamma*b)); one multi-tensor launch each way.  ``red`` is the
    GradReducer that owns the scales' gradients (or None): when it is armed the backward pass accumulates
    straight into its bucket views, as the network passes do (parallel.GradReducer)."""

    @staticmethod
    def forward(ctx, gamma0, signs, red, *gammas):
        gs = [g.detach() for g in gammas]
        ctx.gamma0, ctx.signs, ctx.gs, ctx.red, ctx.params = gamma0, signs, gs, red, gammas
        return ops.sign_loss_fwd(gs, signs, gamma0)

    @staticmethod
    def backward(ctx, gout):
        red = ctx.red
        if red is not None:
            red.begin_pass()
            if red.armed and all(red.owns(p) for p in ctx.params):
                ops.sign_loss_bwd(ctx.gs, ctx.signs, ctx.gamma0, gout.contiguous(),
                                  outs=[red.view_of(p) for p in ctx.params], beta=1.0)
                for p in ctx.params:
                    red.touch(p)
                red.end_pass()
                return (None, None, None) + (None,) * len(ctx.params)
            red.end_pass()
        grads = ops.sign_loss_bwd(ctx.gs, ctx.signs, ctx.gamma0, gout.contiguous())
        return (None, None, None, *grads)


class SignLossModel(nn.Module):
    def __init__(self, model, config, **kwargs):
        super().__init__()
        self.gamma_0 = config.gamma_0
        self.bit_gen = BitGenerator(config.string)
        self._create_signs(model)

    def _create_signs(self, model):
        # gamma <- |gamma| * b in place, one +-1 buffer per norm layer named after the module path
        for safe, m in _norm_layers(model):
            sign = torch.tensor(self.bit_gen.get(m.weight.size(0)), dtype=torch.float32) * 2 - 1
            m.weight.data.abs_().mul_(sign.to(m.weight.data.device))
            self.register_buffer(safe, sign)

    def _pairs(self, model):
        gammas, signs = [], []
        for safe, m in _norm_layers(model):
            gammas.append(m.weight)
            signs.append(getattr(self, safe))
        return gammas, signs

    def forward(self, model):
        gammas, signs = self._pairs(model)
        red = None
        if torch.is_grad_enabled() and gammas and gammas[0].requires_grad:
            from . import parallel
            red = parallel.owner_of(gammas[0])
            if red is not None:
                red.note_forward()
        return _SignLossFn.apply(self.gamma_0, signs, red, *gammas)

    def compute_ber(self, model):
        """bit errors / bits; the count is an exact int64 device reduction (sign(0) is an error)."""
        gammas, signs = self._pairs(model)
        counts = ops.sign_ber_counts([g.detach() for g in gammas], signs)
        # tensor / tensor is an IEEE division on the GPU (tensor / python-scalar multiplies by a
        # rounded reciprocal and can differ from the reference's CPU result in the last bit)
        return counts[0].to(torch.float32) / counts[1].to(torch.float32)


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, x, y):
        xd = x.detach().contiguous()
        yd = None if y is None else y.detach().contiguous()
        ctx.kind, ctx.x, ctx.y = kind, xd, yd
        return ops.loss_fwd(kind, xd, yd)

    @staticmethod
    def backward(ctx, gout):
        return None, ops.loss_bwd(ctx.kind, ctx.x, ctx.y, gout.contiguous()), None


class _LossSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, scale, x, y):
        xd = x.detach().contiguous()
        yd = None if y is None else y.detach().contiguous()
        ctx.kind, ctx.scale, ctx.x, ctx.y = kind, scale, xd, yd
        return ops.loss_sum_fwd(kind, xd, yd, scale)

    @staticmethod
    def backward(ctx, gout):
        return None, None, ops.loss_sum_bwd(ctx.kind, ctx.x, ctx.y, gout.contiguous(), ctx.scale), None


def loss_sum(kind, x, y=None, scale=1.0):
    """``scale * sum(term)`` (reduction='sum' / N of models/vae.py:36-48); gradient flows to ``x`` only."""
    return _LossSumFn.apply(kind, float(scale), x, y)


class _LossPairFn(torch.autograd.Function):
    """(loss_a over the first half, loss_b over the second, their sum) of one vector; the gradient flows from the SUM only."""

    @staticmethod
    def forward(ctx, kind_a, kind_b, x, n_half):
        xd = x.detach().contiguous()
        ctx.kinds, ctx.x, ctx.n = (kind_a, kind_b), xd, n_half
        out = ops.loss_pair_fwd(kind_a, kind_b, xd, n_half)
        la, lb, ls = out[0], out[1], out[2]
        ctx.mark_non_differentiable(la, lb)
        return la, lb, ls

    @staticmethod
    def backward(ctx, ga, gb, gs):
        return None, None, ops.loss_pair_bwd(ctx.kinds[0], ctx.kinds[1], ctx.x, ctx.n, gs.contiguous()), None


PAIR_LOSS_MAX = 256


def loss_pair(kind_a, kind_b, x, n_half):
    """The mean losses of the two halves of ``x`` ([2 * n_half], n_half <= PAIR_LOSS_MAX) and their sum, one launch each way
    (include/iprgan.h: iprgan_loss_pair_*); bit-identical to two ``loss_value`` calls on the halves and an add."""
    return _LossPairFn.apply(kind_a, kind_b, x, int(n_half))


def loss_value(kind, x, y=None):
    """Mean-reduced loss of ``kind`` (include/iprgan.h IPRGAN_LOSS_*); gradient flows to ``x`` only."""
    return _LossFn.apply(kind, x, y)


class Loss(object):
    """tools/loss.py:10-20: optional (x+1)/2 de-normalisation of both arguments, then the mean-reduced fn.  The
    de-normalisation runs inside the loss kernel (IPRGAN_LOSS_*_DENORM) with the reference's own rounding."""

    def __init__(self, kind, normalized=False):
        self.kind, self.denorm = kind, normalized

    def __call__(self, x, y):
        kind = self.kind
        if self.denorm:
            kind = {L.LOSS_L1: L.LOSS_L1_DENORM, L.LOSS_MSE: L.LOSS_MSE_DENORM}[kind]
        return loss_value(kind, x, y)


def l1(normalized=False):
    return Loss(L.LOSS_L1, normalized=normalized)


def mse(normalized=False):
    return Loss(L.LOSS_MSE, normalized=normalized)


class _SSIMFn(torch.autograd.Function):
    """1 - mean SSIM (include/iprgan.h iprgan_ssim_*); the gradient flows to ``x`` only."""

    @staticmethod
    def forward(ctx, x, y, denorm):
        xd, yd = x.detach().contiguous(), y.detach().contiguous()
        want = x.requires_grad
        out, gm = ops.ssim_fwd(xd, yd, denorm, want)
        ctx.x, ctx.y, ctx.gm, ctx.denorm = xd, yd, gm, denorm
        return out

    @staticmethod
    def backward(ctx, gout):
        return ops.ssim_bwd(ctx.x, ctx.y, ctx.gm, gout.contiguous(), ctx.denorm), None, None


class _SSIMLoss(object):
    """tools/loss.py:82-85: ``Loss(lambda x, y: 1 - SSIM(data_range=1)(x, y), normalized)``; the (x+1)/2
    de-normalisation of tools/loss.py:15-18 happens inside the kernel."""

    def __init__(self, normalized=False):
        self.denorm = normalized

    def __call__(self, x, y):
        return _SSIMFn.apply(x, y, bool(self.denorm))


def ssim(normalized=False):
    return _SSIMLoss(normalized=normalized)


class _MSSSIMFn(torch.autograd.Function):
    """1 - MS-SSIM (include/iprgan.h iprgan_msssim_*); the gradient flows to ``x`` only."""

    @staticmethod
    def forward(ctx, x, y, denorm):
        xd, yd = x.detach().contiguous(), y.detach().contiguous()
        out, state = ops.msssim_fwd(xd, yd, denorm, x.requires_grad)
        ctx.x, ctx.y, ctx.state, ctx.denorm = xd, yd, state, denorm
        return out

    @staticmethod
    def backward(ctx, gout):
        return ops.msssim_bwd(ctx.x, ctx.y, ctx.state, gout.contiguous(), ctx.denorm), None, None


class _MSSSIMLoss(object):
    """tools/loss.py:78-80: ``Loss(lambda x, y: 1 - MS_SSIM(data_range=1)(x, y), normalized)`` on the HIP kernels
    (five scales; images must be larger than 160 pixels on their smaller side, as in pytorch-msssim)."""

    def __init__(self, normalized=False):
        self.denorm = normalized

    def __call__(self, x, y):
        return _MSSSIMFn.apply(x, y, bool(self.denorm))


def ms_ssim(normalized=False):
    return _MSSSIMLoss(normalized=normalized)


# ---- black-box trigger / target transforms (tools/transform_dist.py, random_bitmask.py, transform_var.py,
# ---- random_noise_patch.py, paste_watermark.py).  Tiny host-side tensor edits outside the hot path: plain
# ---- torch ops on whatever device the tensors live on.
class TransformDist(nn.Module):
    """z -> sqrt(2 pi) * Phi(z) (tools/transform_dist.py:5-14)."""

    def __init__(self, config=None, **kwargs):
        super().__init__()

    def forward(self, z):
        return 0.5 * (1 + torch.erf(z / math.sqrt(2))) * math.sqrt(2 * math.pi)

    def reset(self): pass


class RandomBitMask(nn.Module):
    """Overwrites ``n_bit`` randomly chosen latent coordinates with ``constant`` (tools/random_bitmask.py:4-29)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.n, self.c, self.z_dim = config.n_bit, config.constant, config.z_dim
        self.reset()

    @torch.no_grad()
    def forward(self, z):
        return z.clone().scatter_(1, self._mask.repeat(z.size(0), 1), self.c)

    def reset(self):
        mask = torch.randperm(self.z_dim)[:self.n].unsqueeze(0)
        if hasattr(self, '_mask'):
            mask = mask.to(self._mask.device)
        self.register_buffer('_mask', mask)

    @property
    def mask(self):
        return self._mask

    @mask.setter
    def mask(self, mask):
        self._mask = mask


class TransformVar(nn.Module):
    """z * (1 - a) + a * w with a ~ Bernoulli(0.25), w = exp(|N(0,1)|) (tools/transform_var.py:5-16)."""

    def __init__(self, config=None, **kwargs):
        super().__init__()
        self.register_buffer('w', torch.ones(1, 128))
        self.register_buffer('a', torch.ones(1, 128))
        self.reset()

    def forward(self, z):
        return z * (1 - self.a) + self.a * self.w

    def reset(self):
        dev = self.w.device
        self.w = torch.exp(torch.randn_like(self.w).abs())
        self.a = (torch.rand(1, 128) < 0.25).float().to(dev)


class _Patch(nn.Module):
    """Shared body of RandomNoisePatch / PasteWatermark: paste ``fg`` where ``bg`` is 0 into one image corner
    (random_noise_patch.py:33-53, paste_watermark.py:42-60)."""

    def _place(self, config, normalized):
        self.position = config.get('position', 'tl')
        assert self.position in ('tl', 'tr', 'bl', 'br'), 'invalid position'
        if normalized:                              # TF.normalize(fg, [0.5]*3, [0.5]*3)
            self.fg = (self.fg - 0.5) / 0.5
        y, x = self.position
        s = config.size
        self.y = (None, s) if y == 't' else (-s, None)
        self.x = (None, s) if x == 'l' else (-s, None)

    @torch.no_grad()
    def forward(self, x):
        (hi, hj), (wi, wj) = self.y, self.x
        y = x.clone()
        y[..., hi:hj, wi:wj] *= self.bg
        y[..., hi:hj, wi:wj] += (1 - self.bg) * self.fg
        return y

    @torch.no_grad()
    def apply_mask(self, x):
        (hi, hj), (wi, wj) = self.y, self.x
        y = torch.ones_like(x[..., hi:hj, wi:wj])
        y *= self.bg
        y += (1 - self.bg) * x[..., hi:hj, wi:wj]
        return y


class RandomNoisePatch(_Patch):
    """Uniform-noise square in a corner (tools/random_noise_patch.py:6-53)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.config, self.normalized = config, kwargs.get('normalized', False)
        self.reset()

    def reset(self):
        size = (self.config.size,) * 2
        dev = self.fg.device if hasattr(self, 'fg') else torch.device('cpu')
        fg = torch.rand(3, *size)
        self.register_buffer('bg', torch.zeros(1, 1, *size))
        self.register_buffer('fg', fg.view(1, 3, *size))
        self._place(self.config, self.normalized)
        self.to(dev)


class PasteWatermark(_Patch):
    """Logo from an RGBA image file, composited on white, resized to ``size`` (tools/paste_watermark.py:6-40);
    PIL only (the reference goes through torchvision's PIL wrappers ``TF.resize`` / ``TF.to_tensor``)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.config, self.normalized = config, kwargs.get('normalized', False)
        import numpy as np
        from PIL import Image
        size = (config.size,) * 2
        tmp = Image.open(config.watermark).convert('RGBA').resize(size, Image.BILINEAR)
        img = Image.new('RGBA', size, 'white')
        img.paste(tmp, (0, 0), mask=tmp)
        to_tensor = lambda im: torch.from_numpy(np.asarray(im, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255
        fg = to_tensor(img.convert('RGB'))
        if config.opaque:
            bg = torch.zeros_like(fg[0:1])
        else:
            mask = Image.new('RGBA', size, (0,) * 4)
            mask.paste(tmp, (0, 0), mask=tmp)
            bg = (to_tensor(mask)[3:] == 0).float()
        self.register_buffer('bg', bg.view(1, 1, *size))
        self.register_buffer('fg', fg.view(1, 3, *size))
        self._place(config, self.normalized)
