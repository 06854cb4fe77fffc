"""Networks of the hot path, same factory names / constructor arguments / state_dict layout as the
reference's ``networks`` package (looked up by string from the YAML config, models/dcgan.py:10-11),
but ``forward`` runs the HIP chain executor (iprgan.engine) instead of ATen/cuDNN modules.

The nn.Module tree is kept only as the owner of parameters and buffers: the leaves are ordinary
``nn.Conv2d`` / ``nn.BatchNorm2d`` / ... objects so that ``state_dict()`` keys, default initialisers,
``spectral_norm``'s ``weight_orig/weight_u/weight_v`` triplets and ``isinstance(m, nn.BatchNorm2d)``
scans (tools/sign_model.py:35, sign_flip.py:61-75) behave exactly as in the reference.  Their own
``forward`` methods are never called by the engine.
"""
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm

from . import _lib as L
from . import engine as E
from .ops import ConvSpec

__all__ = ['ConvGenerator', 'ConvGenerator32', 'ConvGenerator64',
           'SNDiscriminator', 'SNDiscriminator32', 'SNDiscriminator64', 'Flatten']


class _HipNet(nn.Module):
    """Base: builds the chain lazily (after parameters exist) and dispatches forward to it."""

    def _build_chain(self):
        raise NotImplementedError

    def chain(self):
        ch = self.__dict__.get('_chain')
        if ch is None:
            ch = self._build_chain()
            self.__dict__['_chain'] = ch      # not a submodule / not in state_dict
        return ch

    def run(self, x):
        return self.chain()(x, self.training)


# ------------------------------------------------------------------------------------------------
# DCGAN generator: reference networks/conv_generator.py:3-33
# ------------------------------------------------------------------------------------------------
class ConvGenerator(_HipNet):
    CHANNELS = (512, 256, 128, 64)

    def __init__(self, mg, z_dim=128):
        super().__init__()
        self.mg = mg
        ch = self.CHANNELS
        self.fc = nn.Sequential(nn.Linear(z_dim, ch[0] * mg * mg), nn.ReLU(inplace=True))
        up = []
        for cin, cout in zip(ch, ch[1:]):
            up.append(nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=False),
                                    nn.BatchNorm2d(cout), nn.ReLU(inplace=True)))
        up += [nn.ConvTranspose2d(ch[-1], 3, 3, 1, 1, bias=False), nn.Tanh()]
        self.convs = nn.Sequential(*up)

    def _build_chain(self):
        ch, mg = self.CHANNELS, self.mg
        plan = [E.LinearNHWC(self.fc[0], ch[0], mg * mg, act=L.ACT_RELU), E.View((mg, mg, ch[0]))]
        for i, (cin, cout) in enumerate(zip(ch, ch[1:])):
            blk = self.convs[i]
            plan.append(E.Conv(ConvSpec(cin, cout, 4, 2, 1, transposed=True), blk[0]))
            plan.append(E.BatchNorm(blk[1], act=L.ACT_RELU))
        last = self.convs[len(ch) - 1]
        plan.append(E.Conv(ConvSpec(ch[-1], 3, 3, 1, 1, transposed=True, act=L.ACT_TANH), last))
        plan.append(E.ToNCHW(3))
        return E.Chain(plan)

    def forward(self, z):
        return self.run(z)


def ConvGenerator32():
    return ConvGenerator(mg=4)


def ConvGenerator64():
    return ConvGenerator(mg=8)


# ------------------------------------------------------------------------------------------------
# DCGAN spectral-norm discriminator: reference networks/sn_discriminator.py:4-38
# ------------------------------------------------------------------------------------------------
class Flatten(nn.Module):
    """Placeholder keeping the reference's Sequential indices (net.5); the engine folds the NCHW
    flatten order into the GEMV head's weight permutation."""

    def forward(self, x):
        return x.flatten(1)


class SNDiscriminator(_HipNet):
    STAGES = ((3, 64), (64, 128), (128, 256))
    SLOPE = 0.1

    def __init__(self, md):
        super().__init__()
        self.md = md
        act = lambda: nn.LeakyReLU(negative_slope=self.SLOPE, inplace=True)
        mods = []
        for cin, cout in self.STAGES:
            mods.append(nn.Sequential(spectral_norm(nn.Conv2d(cin, cout, 3, 1, 1, bias=True)), act(),
                                      spectral_norm(nn.Conv2d(cout, cout, 4, 2, 1, bias=True)), act()))
        mods += [spectral_norm(nn.Conv2d(256, 512, 3, 1, 1, bias=True)), act(), Flatten(),
                 spectral_norm(nn.Linear(512 * md * md, 1))]
        self.net = nn.Sequential(*mods)

    def _build_chain(self):
        lr = dict(act=L.ACT_LRELU, slope=self.SLOPE)
        plan = [E.ToNHWC(3)]
        for i, (cin, cout) in enumerate(self.STAGES):
            blk = self.net[i]
            plan.append(E.Conv(ConvSpec(cin, cout, 3, 1, 1, **lr), blk[0], sn=True))
            plan.append(E.Conv(ConvSpec(cout, cout, 4, 2, 1, **lr), blk[2], sn=True))
        n = len(self.STAGES)
        plan.append(E.Conv(ConvSpec(256, 512, 3, 1, 1, **lr), self.net[n], sn=True))
        plan.append(E.GemvHead(self.net[n + 3], 512, self.md * self.md))
        return E.Chain(plan)

    def forward(self, x):
        return self.run(x).view(-1)


def SNDiscriminator32():
    return SNDiscriminator(md=4)


def SNDiscriminator64():
    return SNDiscriminator(md=8)
