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

import os

from . import _lib as L
from . import engine as E
from . import ops
from .ops import ConvSpec, reparam_bwd as ops_reparam_bwd, reparam_fwd as ops_reparam_fwd

_FUSE_PRELU = os.environ.get('IPRGAN_FUSE_PRELU', '1') != '0'      # A/B switch: PReLU folded into the BatchNorm in front of it

__all__ = ['ConvGenerator', 'ConvGenerator32', 'ConvGenerator64', 'ConvGenerator128',
           'SNDiscriminator', 'SNDiscriminator32', 'SNDiscriminator64', 'SNDiscriminator128', 'Flatten',
           'SRResNet', 'Discriminator96', 'VGG19Feature',
           'ResnetGenerator', 'ResnetBlock', 'Resnet6Blocks', 'Resnet9Blocks', 'ConvDiscriminator',
           'Encoder32', 'Decoder32']


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


def ConvGenerator128():
    """Not in the reference (BASELINE config 5 'DCGAN 128x128': SURVEY section 8a constructs ConvGenerator(mg=16))."""
    return ConvGenerator(mg=16)


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

    def forward_pair(self, xa, xb):
        """D(xa), D(xb) of two equally sized batches in ONE pass of twice the batch (not reference API: the engine's
        form of the reference's two consecutive calls, models/dcgan.py:47-48).  The spectral-norm power iteration runs
        twice, as for two calls, and each half is normalised by its own sigma; the buffers end where two calls leave
        them.  Launches of twice the size fill the GPU better (+5-10 % per layer at batch 128 + 128)."""
        out = self.forward_pair_flat(xa, xb)
        n = xa.shape[0]
        return out[:n], out[n:]

    def forward_pair_flat(self, xa, xb):
        """forward_pair's two results as ONE [2B] tensor (first half D(xa)): what a fused pair loss consumes without autograd's
        slice gradients (tools.loss_pair)."""
        return self.chain()(torch.cat([xa, xb]), self.training, pair=True).view(-1)

    @staticmethod
    def can_pair(x):
        """The largest activation of the paired pass ([2B, H, W, 64] behind the first convolution: 4 bytes per element as
        fp32, 6 as three planes inside one buffer range) must stay below the 2 GiB tensor limit of the convolution entry
        points (include/iprgan.h).  bf16 storage is held to the fp32 size: DCGAN-128 at batch 256 + 256 would fit as bf16 and
        was measured in round 6 - 16.75 / 16.59 ms paired against 16.69 / 16.64 in two passes: layers of that size fill the
        GPU either way."""
        B, _, H, W = x.shape
        bpe = 6 if ops.act_kind(64) == ops.ST_X3 else 4
        return 2 * B * H * W * 64 * bpe < (1 << 31)


def SNDiscriminator32():
    return SNDiscriminator(md=4)


def SNDiscriminator64():
    return SNDiscriminator(md=8)


def SNDiscriminator128():
    """Not in the reference (BASELINE config 5): SNDiscriminator(md=16)."""
    return SNDiscriminator(md=16)


# ------------------------------------------------------------------------------------------------
# SRGAN generator: reference networks/sr_resnet.py:3-45
# ------------------------------------------------------------------------------------------------
def _kaiming(conv, a):
    nn.init.kaiming_normal_(conv.weight.data, a=a, mode='fan_in')
    conv.bias.data.zero_()
    return conv


def _sr_unit(cin, cout, k, pad, norm=False, prelu=False):
    """conv [+BN] [+PReLU] as one Sequential; kaiming_normal(a=.25 with an activation, else 1)."""
    mods = [nn.Conv2d(cin, cout, k, 1, pad)]
    if norm:
        mods.append(nn.BatchNorm2d(cout))
    if prelu:
        mods.append(nn.PReLU())
    _kaiming(mods[0], 0.25 if prelu else 1.0)
    return nn.Sequential(*mods)


class _Residual(nn.Module):
    def __init__(self, block):
        super().__init__()
        self.block = block

    def forward(self, x):                   # never used by the engine
        return x + self.block(x)


class SRResNet(nn.Sequential, _HipNet):
    def __init__(self, n_block=16):
        trunk = [_Residual(nn.Sequential(_sr_unit(64, 64, 3, 1, norm=True, prelu=True),
                                         _sr_unit(64, 64, 3, 1, norm=True)))
                 for _ in range(n_block)]
        trunk.append(_sr_unit(64, 64, 3, 1, norm=True))
        head = _sr_unit(3, 64, 9, 4, prelu=True)
        body = _Residual(nn.Sequential(*trunk))
        ups = []
        for _ in range(2):
            ups.append(nn.Sequential(_sr_unit(64, 256, 3, 1), nn.PixelShuffle(2), nn.PReLU()))
        tail = _sr_unit(64, 3, 9, 4)
        nn.Sequential.__init__(self, head, body, *ups, tail)
        self.n_block = n_block

    def _unit_ops(self, unit, cin, cout, k, pad):
        ops_ = [E.Conv(ConvSpec(cin, cout, k, 1, pad), unit[0])]
        rest = list(unit)[1:]
        if _FUSE_PRELU and len(rest) == 2 and isinstance(rest[0], nn.BatchNorm2d) and isinstance(rest[1], nn.PReLU):
            return ops_ + [E.BatchNorm(rest[0], prelu=rest[1])]       # conv -> BatchNorm -> PReLU: PReLU rides on the norm's passes
        for m in rest:
            ops_.append(E.BatchNorm(m) if isinstance(m, nn.BatchNorm2d) else E.PReLU(m))
        return ops_

    def _build_chain(self):
        plan = [E.ToNHWC(3)] + self._unit_ops(self[0], 3, 64, 9, 4)
        plan.append(E.SkipStart())
        trunk = self[1].block
        for i in range(self.n_block):
            plan.append(E.SkipStart())
            plan += self._unit_ops(trunk[i].block[0], 64, 64, 3, 1)
            plan += self._unit_ops(trunk[i].block[1], 64, 64, 3, 1)
            plan.append(E.SkipEnd())
        plan += self._unit_ops(trunk[self.n_block], 64, 64, 3, 1)
        plan.append(E.SkipEnd())
        for up in (self[2], self[3]):
            plan += self._unit_ops(up[0], 64, 256, 3, 1)
            # PixelShuffle -> PReLU (one slope: it commutes with the permutation) in one pass each way
            plan += [E.PixelShufflePReLU(up[2])] if _FUSE_PRELU else [E.PixelShuffle2(), E.PReLU(up[2])]
        plan += self._unit_ops(self[4], 64, 3, 9, 4)
        plan.append(E.ToNCHW(3))
        return E.Chain(plan)

    def forward(self, x):
        return self.run(x)


# ------------------------------------------------------------------------------------------------
# SRGAN discriminator: reference networks/discriminator_96.py:3-35
# ------------------------------------------------------------------------------------------------
class Discriminator96(nn.Sequential, _HipNet):
    PLAN = ((64, 64, 2), (64, 128, 1), (128, 128, 2), (128, 256, 1), (256, 256, 2), (256, 512, 1), (512, 512, 2))
    SLOPE = 0.2

    def __init__(self):
        mods = [nn.Conv2d(3, 64, 3, 1, 1), nn.LeakyReLU(self.SLOPE, True)]
        for cin, cout, s in self.PLAN:
            blk = nn.Sequential(nn.Conv2d(cin, cout, 3, s, 1), nn.BatchNorm2d(cout), nn.LeakyReLU(self.SLOPE, True))
            _kaiming(blk[0], 0.2)
            mods.append(blk)
        mods += [nn.Conv2d(512, 1024, 6, 1, 0), nn.LeakyReLU(self.SLOPE, True), nn.Conv2d(1024, 1, 1, 1, 0)]
        nn.Sequential.__init__(self, *mods)

    def _build_chain(self):
        lr = dict(act=L.ACT_LRELU, slope=self.SLOPE)
        plan = [E.ToNHWC(3), E.Conv(ConvSpec(3, 64, 3, 1, 1, **lr), self[0])]
        for i, (cin, cout, s) in enumerate(self.PLAN):
            blk = self[2 + i]
            plan.append(E.Conv(ConvSpec(cin, cout, 3, s, 1), blk[0]))
            plan.append(E.BatchNorm(blk[1], **lr))
        plan.append(E.Conv(ConvSpec(512, 1024, 6, 1, 0, **lr), self[9]))
        plan.append(E.Conv(ConvSpec(1024, 1, 1, 1, 0), self[11]))
        plan.append(E.Squeeze())
        return E.Chain(plan)

    def forward(self, x):
        return self.run(x).squeeze()


# ------------------------------------------------------------------------------------------------
# VGG19 feature extractor: reference networks/vgg.py:5-40 (architecture = torchvision vgg19 cfg "E";
# pretrained weights cannot be downloaded here: random init unless a state_dict is loaded)
# ------------------------------------------------------------------------------------------------
class VGG19Feature(_HipNet):
    CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M')

    def __init__(self, layer='relu5_4'):
        super().__init__()
        names, mods, cin, blk, idx = [], [], 3, 1, 1
        for v in self.CFG:
            if v == 'M':
                names.append(f'pool{blk}')
                mods.append(nn.MaxPool2d(kernel_size=2, stride=2))
                blk, idx = blk + 1, 1
            else:
                names += [f'conv{blk}_{idx}', f'relu{blk}_{idx}']
                mods += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin, idx = v, idx + 1
        self.net = nn.Sequential(*mods[:names.index(layer) + 1])
        self.net.eval()
        for p in self.parameters():
            p.requires_grad = False
        # The reference downloads torchvision's ImageNet weights (networks/vgg.py:33: vgg19(pretrained=True)); a box without
        # network access gets them from a file: IPRGAN_VGG19_WEIGHTS=<vgg19-*.pth> (torchvision's own checkpoint layout).
        path = os.environ.get('IPRGAN_VGG19_WEIGHTS')
        if path:
            self.load_torchvision_state_dict(torch.load(path, map_location='cpu'))

    def load_torchvision_state_dict(self, sd):
        """Weights in torchvision's vgg19 layout (``features.<i>.weight / .bias`` with i the index inside ``features``;
        ``classifier.*`` and the feature layers past this extractor's cut are ignored) -> ``self.net``, whose indices are
        the same ones (networks/vgg.py:33 slices ``vgg19().features``).  Every convolution of the extractor must be covered."""
        own = self.net.state_dict()
        got = {k[len('features.'):]: v for k, v in sd.items() if k.startswith('features.') and k[len('features.'):] in own}
        missing = sorted(set(own) - set(got))
        if missing:
            raise KeyError(f'VGG19Feature: the state_dict lacks features.{{{", ".join(missing[:4])}, ...}} ({len(missing)} of {len(own)} tensors)')
        self.net.load_state_dict(got, strict=True)
        return self

    def _build_chain(self):
        plan = [E.ToNHWC(3)]
        mods = list(self.net)
        i, cin = 0, 3
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                fused = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                plan.append(E.Conv(ConvSpec(cin, m.out_channels, 3, 1, 1, act=L.ACT_RELU if fused else L.ACT_NONE), m))
                cin = m.out_channels
                i += 2 if fused else 1
            elif isinstance(m, nn.MaxPool2d):
                plan.append(E.MaxPool2())
                i += 1
            else:
                raise RuntimeError(f'unexpected module {m} in VGG19Feature')
        plan.append(E.ToNCHW(cin))
        return E.Chain(plan)

    def forward(self, x):
        return self.run(x)


# ------------------------------------------------------------------------------------------------
# CycleGAN generator: reference networks/resnet_generator.py:3-59
# ------------------------------------------------------------------------------------------------
class ResnetBlock(nn.Module):
    def __init__(self, channel):
        super().__init__()
        seq = []
        for last in (False, True):
            seq += [nn.ReflectionPad2d(1), nn.Conv2d(channel, channel, 3, 1, 0, bias=True),
                    nn.InstanceNorm2d(channel, affine=True)]
            if not last:
                seq.append(nn.ReLU(True))
        self.block = nn.Sequential(*seq)

    def forward(self, x):                   # never used by the engine
        return x + self.block(x)


class ResnetGenerator(nn.Sequential, _HipNet):
    def __init__(self, n_block):
        mods = [nn.ReflectionPad2d(3), nn.Conv2d(3, 64, 7, 1, 0), nn.InstanceNorm2d(64, affine=True), nn.ReLU(True)]
        for c in (64, 128):
            mods += [nn.Conv2d(c, 2 * c, 3, 2, 1), nn.InstanceNorm2d(2 * c, affine=True), nn.ReLU(True)]
        mods += [ResnetBlock(256) for _ in range(n_block)]
        for c in (256, 128):
            mods += [nn.ConvTranspose2d(c, c // 2, 3, 2, 1, output_padding=1),
                     nn.InstanceNorm2d(c // 2, affine=True), nn.ReLU(True)]
        mods += [nn.ReflectionPad2d(3), nn.Conv2d(64, 3, 7, 1, 0), nn.Tanh()]
        nn.Sequential.__init__(self, *mods)
        self.n_block = n_block

    def _build_chain(self):
        R = L.PAD_REFLECT
        plan = [E.ToNHWC(3),
                E.Conv(ConvSpec(3, 64, 7, 1, 3, pad_mode=R), self[1]), E.InstanceNorm(self[2], act=L.ACT_RELU)]
        i = 4
        for c in (64, 128):
            plan += [E.Conv(ConvSpec(c, 2 * c, 3, 2, 1), self[i]), E.InstanceNorm(self[i + 1], act=L.ACT_RELU)]
            i += 3
        for _ in range(self.n_block):
            b = self[i].block
            plan += [E.SkipStart(),
                     E.Conv(ConvSpec(256, 256, 3, 1, 1, pad_mode=R), b[1]), E.InstanceNorm(b[2], act=L.ACT_RELU),
                     E.Conv(ConvSpec(256, 256, 3, 1, 1, pad_mode=R), b[5]), E.InstanceNorm(b[6]),
                     E.SkipEnd()]
            i += 1
        for c in (256, 128):
            plan += [E.Conv(ConvSpec(c, c // 2, 3, 2, 1, outpad=1, transposed=True), self[i]),
                     E.InstanceNorm(self[i + 1], act=L.ACT_RELU)]
            i += 3
        plan += [E.Conv(ConvSpec(64, 3, 7, 1, 3, pad_mode=R, act=L.ACT_TANH), self[i + 1]), E.ToNCHW(3)]
        return E.Chain(plan)

    def forward(self, x):
        return self.run(x)


def Resnet9Blocks():
    return ResnetGenerator(n_block=9)


def Resnet6Blocks():
    return ResnetGenerator(n_block=6)


# ------------------------------------------------------------------------------------------------
# CycleGAN PatchGAN discriminator: reference networks/conv_discriminator.py:3-22
# ------------------------------------------------------------------------------------------------
class ConvDiscriminator(nn.Sequential, _HipNet):
    SLOPE = 0.2

    def __init__(self):
        act = lambda: nn.LeakyReLU(self.SLOPE, True)
        nn.Sequential.__init__(
            self,
            nn.Conv2d(3, 64, 4, 2, 1), act(),
            nn.Conv2d(64, 128, 4, 2, 1), nn.InstanceNorm2d(128), act(),
            nn.Conv2d(128, 256, 4, 2, 1), nn.InstanceNorm2d(256), act(),
            nn.Conv2d(256, 512, 4, 1, 1), nn.InstanceNorm2d(512), act(),
            nn.Conv2d(512, 1, 4, 1, 1))

    def _build_chain(self):
        lr = dict(act=L.ACT_LRELU, slope=self.SLOPE)
        plan = [E.ToNHWC(3), E.Conv(ConvSpec(3, 64, 4, 2, 1, **lr), self[0])]
        for idx, cin, cout, s in ((2, 64, 128, 2), (5, 128, 256, 2), (8, 256, 512, 1)):
            plan += [E.Conv(ConvSpec(cin, cout, 4, s, 1), self[idx]), E.InstanceNorm(self[idx + 1], **lr)]
        plan += [E.Conv(ConvSpec(512, 1, 4, 1, 1), self[11]), E.ToNCHW(1)]
        return E.Chain(plan)

    def forward(self, x):
        return self.run(x)


# ------------------------------------------------------------------------------------------------
# VAE on 32x32 images: reference networks/encoder.py:4-30 and networks/decoder.py:3-33
# ------------------------------------------------------------------------------------------------
class _ReparamFn(torch.autograd.Function):
    """z = eps * exp(logvar / 2) + mean (networks/encoder.py:24-28)."""

    @staticmethod
    def forward(ctx, mean, logvar, eps):
        m, lv = mean.detach().contiguous(), logvar.detach().contiguous()
        ctx.save_for_backward(lv, eps)
        return ops_reparam_fwd(m, lv, eps)

    @staticmethod
    def backward(ctx, dz):
        lv, eps = ctx.saved_tensors
        dmean, dlogvar = ops_reparam_bwd(dz.contiguous(), lv, eps)
        return dmean, dlogvar, None


class Encoder32(_HipNet):
    """image -> (z, (mean, logvar)).  ``eps_fn(shape, device)``, when set, supplies the N(0,1) draw of the
    reparameterisation (parity tests replay the reference's draw); otherwise ``torch.randn`` on the device."""

    def __init__(self):
        super().__init__()
        self.encoder = nn.Sequential(
            nn.Conv2d(3, 32, 3, 2, 1), nn.BatchNorm2d(32, affine=True), nn.ReLU(inplace=True),
            nn.Conv2d(32, 64, 3, 2, 1), nn.BatchNorm2d(64, affine=True), nn.ReLU(inplace=True),
            nn.Conv2d(64, 128, 3, 2, 1))
        self.q_mean = nn.Linear(2048, 128)
        self.q_logvar = nn.Linear(2048, 128)
        self.__dict__['eps_fn'] = None

    def _build_chain(self):
        e = self.encoder
        body = E.Chain([E.ToNHWC(3),
                        E.Conv(ConvSpec(3, 32, 3, 2, 1), e[0]), E.BatchNorm(e[1], act=L.ACT_RELU),
                        E.Conv(ConvSpec(32, 64, 3, 2, 1), e[3]), E.BatchNorm(e[4], act=L.ACT_RELU),
                        E.Conv(ConvSpec(64, 128, 3, 2, 1), e[6]),
                        E.ToNCHW(128), E.View((2048,))])         # q.flatten(start_dim=1) in NCHW order
        heads = [E.Chain([E.LinearNHWC(m, 128, 1, act=L.ACT_NONE)]) for m in (self.q_mean, self.q_logvar)]
        return body, heads

    def forward(self, x):
        body, heads = self.chain()
        q = body(x, self.training)
        mean, logvar = heads[0](q, self.training), heads[1](q, self.training)
        fn = self.__dict__.get('eps_fn')
        eps = fn(tuple(mean.shape), mean.device) if fn is not None else torch.randn(mean.shape, device=mean.device)
        z = _ReparamFn.apply(mean, logvar, eps.to(mean.device, torch.float32).contiguous())
        return z, (mean, logvar)


class Decoder32(nn.Sequential, _HipNet):
    """z -> image in [-1,1].  Module indices (state_dict keys 0., 2., 3., 5., 6., 8.) follow the reference."""

    class Reshape(nn.Module):
        def __init__(self, *shape):
            super().__init__()
            self.shape = shape

        def forward(self, x):
            return x.view(-1, *self.shape)

    class Normalize(nn.Module):
        def forward(self, x):
            return x * 2 - 1

    def __init__(self):
        nn.Sequential.__init__(
            self,
            nn.Linear(128, 2048), Decoder32.Reshape(128, 4, 4),
            nn.ConvTranspose2d(128, 64, 4, 2, 1), nn.BatchNorm2d(64, affine=True), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(64, 32, 4, 2, 1), nn.BatchNorm2d(32, affine=True), nn.ReLU(inplace=True),
            nn.ConvTranspose2d(32, 3, 4, 2, 1), nn.Sigmoid(), Decoder32.Normalize())

    def _build_chain(self):
        T = dict(transposed=True)
        return E.Chain([E.LinearNHWC(self[0], 128, 16, act=L.ACT_NONE), E.View((4, 4, 128)),
                        E.Conv(ConvSpec(128, 64, 4, 2, 1, **T), self[2]), E.BatchNorm(self[3], act=L.ACT_RELU),
                        E.Conv(ConvSpec(64, 32, 4, 2, 1, **T), self[5]), E.BatchNorm(self[6], act=L.ACT_RELU),
                        E.Conv(ConvSpec(32, 3, 4, 2, 1, act=L.ACT_SIGMOID_PM1, **T), self[8]),   # Sigmoid + x*2-1
                        E.ToNCHW(3)])

    def forward(self, z):
        return self.run(z)
