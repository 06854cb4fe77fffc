"""Watermark tooling on the hot path: the sign-loss regulariser and image losses, with the reference's
names and call signatures (tools/sign_model.py:6-60, tools/loss.py:10-20,72-76) on HIP kernels."""
import random

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

__all__ = ['BitGenerator', 'SignLossModel', 'Loss', 'l1', 'mse', 'loss_value']

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
    """loss = sum_layers mean(relu(gamma0 - gamma*b)); one multi-tensor launch each way."""

    @staticmethod
    def forward(ctx, gamma0, signs, *gammas):
        gs = [g.detach() for g in gammas]
        ctx.gamma0, ctx.signs, ctx.gs = gamma0, signs, gs
        return ops.sign_loss_fwd(gs, signs, gamma0)

    @staticmethod
    def backward(ctx, gout):
        grads = ops.sign_loss_bwd(ctx.gs, ctx.signs, ctx.gamma0, gout.contiguous())
        return (None, None, *grads)


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
        return _SignLossFn.apply(self.gamma_0, signs, *gammas)

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


def loss_value(kind, x, y=None):
    """Mean-reduced loss of ``kind`` (include/iprgan.h IPRGAN_LOSS_*); gradient flows to ``x`` only."""
    return _LossFn.apply(kind, x, y)


class Loss(object):
    """tools/loss.py:10-20: optional (x+1)/2 de-normalisation of both arguments, then fn."""

    def __init__(self, kind, normalized=False):
        self.kind, self.denorm = kind, normalized

    def __call__(self, x, y):
        v = loss_value(self.kind, x, y)
        if self.denorm:
            # L1 scales by 1/2 and MSE by 1/4 under x -> (x+1)/2 on both arguments
            v = v * (0.5 if self.kind == L.LOSS_L1 else 0.25)
        return v


def l1(normalized=False):
    return Loss(L.LOSS_L1, normalized=normalized)


def mse(normalized=False):
    return Loss(L.LOSS_MSE, normalized=normalized)
