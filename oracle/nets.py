"""Oracle networks: plain torch.nn CPU restatements of the seven hot-path nets.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Each builder reproduces the
module tree of the reference so that ``state_dict()`` keys, shapes and the
order of ``named_modules()`` (which fixes the sign-loss bit order,
tools/sign_model.py:33-40) are identical; the layer recipes are tables, the
arithmetic is torch's.
"""
import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm


def _seq(*mods):
    return nn.Sequential(*mods)


# --------------------------------------------------------------------------
# DCGAN generator  (reference networks/conv_generator.py:3-33)
# --------------------------------------------------------------------------
class ConvGenerator(nn.Module):
    """z(128) -> Linear+ReLU -> view(512,mg,mg) -> 3x[ConvT k4s2p1, BN, ReLU]
    -> ConvT k3s1p1 -> Tanh   (conv_generator.py:7-27)."""
    widths = (512, 256, 128, 64)

    def __init__(self, mg, z_dim=128):
        super().__init__()
        self.mg = mg
        w = self.widths
        self.fc = _seq(nn.Linear(z_dim, w[0] * mg * mg), nn.ReLU(inplace=True))
        stages = [
            _seq(nn.ConvTranspose2d(a, b, 4, 2, 1, bias=False),
                 nn.BatchNorm2d(b), nn.ReLU(inplace=True))
            for a, b in zip(w[:-1], w[1:])
        ]
        self.convs = _seq(*stages,
                          nn.ConvTranspose2d(w[-1], 3, 3, 1, 1, bias=False),
                          nn.Tanh())

    def forward(self, z):
        h = self.fc(z)
        return self.convs(h.view(z.size(0), -1, self.mg, self.mg))


def ConvGenerator32():
    return ConvGenerator(mg=4)


def ConvGenerator64():
    return ConvGenerator(mg=8)


def ConvGenerator128():                         # not a reference factory: BASELINE config 5
    return ConvGenerator(mg=16)


# --------------------------------------------------------------------------
# DCGAN spectral-norm discriminator (reference networks/sn_discriminator.py:4-38)
# --------------------------------------------------------------------------
class Flatten(nn.Module):
    def forward(self, x):                       # sn_discriminator.py:27-32
        return x.flatten(1)


class SNDiscriminator(nn.Module):
    """3x[SN conv3 s1, LReLU .1, SN conv4 s2, LReLU .1] -> SN conv3 256->512
    -> LReLU -> flatten -> SN Linear -> view(-1)   (sn_discriminator.py:8-25)."""

    def __init__(self, md):
        super().__init__()

        def pair(a, b):
            return _seq(spectral_norm(nn.Conv2d(a, b, 3, 1, 1)), nn.LeakyReLU(0.1, inplace=True),
                        spectral_norm(nn.Conv2d(b, b, 4, 2, 1)), nn.LeakyReLU(0.1, inplace=True))

        self.net = _seq(pair(3, 64), pair(64, 128), pair(128, 256),
                        spectral_norm(nn.Conv2d(256, 512, 3, 1, 1)),
                        nn.LeakyReLU(0.1, inplace=True),
                        Flatten(),
                        spectral_norm(nn.Linear(512 * md * md, 1)))

    def forward(self, x):
        return self.net(x).view(-1)


def SNDiscriminator32():
    return SNDiscriminator(md=4)


def SNDiscriminator64():
    return SNDiscriminator(md=8)


def SNDiscriminator128():                       # not a reference factory: BASELINE config 5
    return SNDiscriminator(md=16)


# --------------------------------------------------------------------------
# SRGAN generator (reference networks/sr_resnet.py:3-45)
# --------------------------------------------------------------------------
class _SRConv(nn.Sequential):
    """conv [+BN] [+act]; kaiming_normal(a=.25 if act else 1) + zero bias
    (sr_resnet.py:19-29)."""

    def __init__(self, cin, cout, k, s=1, p=0, n=False, a=None):
        layers = [nn.Conv2d(cin, cout, k, s, p)]
        if n:
            layers.append(nn.BatchNorm2d(cout))
        if a is not None:
            layers.append(a)
        super().__init__(*layers)
        nn.init.kaiming_normal_(self[0].weight.data, a=0.25 if a is not None else 1.0, mode='fan_in')
        self[0].bias.data.zero_()


class _Skip(nn.Module):
    def __init__(self, block):                  # sr_resnet.py:31-37
        super().__init__()
        self.block = block

    def forward(self, x):
        return x + self.block(x)


class _Up(nn.Sequential):
    def __init__(self, cin, cout):              # sr_resnet.py:39-45
        super().__init__(_SRConv(cin, cout * 4, 3, 1, 1), nn.PixelShuffle(2), nn.PReLU())


class SRResNet(nn.Sequential):
    def __init__(self, n_block=16):             # sr_resnet.py:4-17
        body = [_Skip(_seq(_SRConv(64, 64, 3, 1, 1, n=True, a=nn.PReLU()),
                           _SRConv(64, 64, 3, 1, 1, n=True)))
                for _ in range(n_block)]
        body.append(_SRConv(64, 64, 3, 1, 1, n=True))
        super().__init__(_SRConv(3, 64, 9, 1, 4, a=nn.PReLU()),
                         _Skip(_seq(*body)),
                         _Up(64, 64), _Up(64, 64),
                         _SRConv(64, 3, 9, 1, 4))


# --------------------------------------------------------------------------
# SRGAN discriminator (reference networks/discriminator_96.py:3-35)
# --------------------------------------------------------------------------
class _D96Block(nn.Sequential):
    def __init__(self, cin, cout, k, s=1, p=0):  # discriminator_96.py:27-35
        super().__init__(nn.Conv2d(cin, cout, k, s, p), nn.BatchNorm2d(cout),
                         nn.LeakyReLU(0.2, True))
        nn.init.kaiming_normal_(self[0].weight.data, a=0.2, mode='fan_in')
        self[0].bias.data.zero_()


class Discriminator96(nn.Sequential):
    def __init__(self):                          # discriminator_96.py:4-22
        chain = [(64, 64, 2), (64, 128, 1), (128, 128, 2), (128, 256, 1),
                 (256, 256, 2), (256, 512, 1), (512, 512, 2)]
        super().__init__(nn.Conv2d(3, 64, 3, 1, 1), nn.LeakyReLU(0.2, True),
                         *[_D96Block(a, b, 3, s, 1) for a, b, s in chain],
                         nn.Conv2d(512, 1024, 6, 1, 0), nn.LeakyReLU(0.2, True),
                         nn.Conv2d(1024, 1, 1, 1, 0))

    def forward(self, x):                        # discriminator_96.py:24-25
        return super().forward(x).squeeze()


# --------------------------------------------------------------------------
# VGG19 features[:36]  (reference networks/vgg.py:5-40; the architecture lives in
# third-party torchvision 0.9.0 `vgg19().features`, cfg "E"; restated here.
# Pretrained weights are unavailable offline -> parity unpinned, random init.)
# --------------------------------------------------------------------------
_VGG_E = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
          512, 512, 512, 512, 'M', 512, 512, 512, 512, 'M']


class VGG19Feature(nn.Module):
    def __init__(self, layer='relu5_4'):
        super().__init__()
        names, mods, cin = [], [], 3
        blk, idx = 1, 1
        for v in _VGG_E:
            if v == 'M':
                names.append(f'pool{blk}')
                mods.append(nn.MaxPool2d(2, 2))
                blk, idx = blk + 1, 1
            else:
                names += [f'conv{blk}_{idx}', f'relu{blk}_{idx}']
                mods += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin, idx = v, idx + 1
        cut = names.index(layer) + 1            # vgg.py:32-33
        self.net = nn.Sequential(*mods[:cut])
        self.net.eval()
        for p in self.parameters():             # vgg.py:36-37
            p.requires_grad = False

    def forward(self, x):
        return self.net(x)


# --------------------------------------------------------------------------
# CycleGAN generator (reference networks/resnet_generator.py:3-59)
# --------------------------------------------------------------------------
class ResnetBlock(nn.Module):
    def __init__(self, ch):                      # resnet_generator.py:39-53
        super().__init__()
        self.block = _seq(nn.ReflectionPad2d(1), nn.Conv2d(ch, ch, 3, 1, 0, bias=True),
                          nn.InstanceNorm2d(ch, affine=True), nn.ReLU(True),
                          nn.ReflectionPad2d(1), nn.Conv2d(ch, ch, 3, 1, 0, bias=True),
                          nn.InstanceNorm2d(ch, affine=True))

    def forward(self, x):
        return x + self.block(x)


class ResnetGenerator(nn.Sequential):
    def __init__(self, n_block):                 # resnet_generator.py:4-37
        layers = [nn.ReflectionPad2d(3), nn.Conv2d(3, 64, 7, 1, 0),
                  nn.InstanceNorm2d(64, affine=True), nn.ReLU(True)]
        for ch in (64, 128):
            layers += [nn.Conv2d(ch, ch * 2, 3, 2, 1),
                       nn.InstanceNorm2d(ch * 2, affine=True), nn.ReLU(True)]
        layers += [ResnetBlock(256) for _ in range(n_block)]
        for ch in (256, 128):
            layers += [nn.ConvTranspose2d(ch, ch // 2, 3, 2, 1, output_padding=1),
                       nn.InstanceNorm2d(ch // 2, affine=True), nn.ReLU(True)]
        layers += [nn.ReflectionPad2d(3), nn.Conv2d(64, 3, 7, 1, 0), nn.Tanh()]
        super().__init__(*layers)


def Resnet9Blocks():
    return ResnetGenerator(n_block=9)


def Resnet6Blocks():
    return ResnetGenerator(n_block=6)


# --------------------------------------------------------------------------
# CycleGAN PatchGAN discriminator (reference networks/conv_discriminator.py:3-22)
# --------------------------------------------------------------------------
class ConvDiscriminator(nn.Sequential):
    def __init__(self):
        super().__init__(
            nn.Conv2d(3, 64, 4, 2, 1), nn.LeakyReLU(0.2, True),
            nn.Conv2d(64, 128, 4, 2, 1), nn.InstanceNorm2d(128), nn.LeakyReLU(0.2, True),
            nn.Conv2d(128, 256, 4, 2, 1), nn.InstanceNorm2d(256), nn.LeakyReLU(0.2, True),
            nn.Conv2d(256, 512, 4, 1, 1), nn.InstanceNorm2d(512), nn.LeakyReLU(0.2, True),
            nn.Conv2d(512, 1, 4, 1, 1))


# --------------------------------------------------------------------------
# VAE encoder / decoder on 32x32 (reference networks/encoder.py:4-30, networks/decoder.py:3-33)
# --------------------------------------------------------------------------
class Encoder32(nn.Module):
    """3x Conv k3s2p1 (BN+ReLU after the first two) -> flatten(2048) -> two Linear(2048,128) heads;
    z = eps * exp(logvar/2) + mean with eps = randn_like (encoder.py:20-30)."""

    def __init__(self):
        super().__init__()
        self.encoder = _seq(nn.Conv2d(3, 32, 3, 2, 1), nn.BatchNorm2d(32, affine=True), nn.ReLU(inplace=True),
                            nn.Conv2d(32, 64, 3, 2, 1), nn.BatchNorm2d(64, affine=True), nn.ReLU(inplace=True),
                            nn.Conv2d(64, 128, 3, 2, 1))
        self.q_mean = nn.Linear(2048, 128)
        self.q_logvar = nn.Linear(2048, 128)

    def forward(self, x):
        q = self.encoder(x).flatten(start_dim=1)
        mean, logvar = self.q_mean(q), self.q_logvar(q)
        std = torch.exp(logvar * 0.5)
        eps = torch.randn_like(std)
        return eps * std + mean, (mean, logvar)


class _Reshape(nn.Module):
    def __init__(self, *shape):
        super().__init__()
        self.shape = shape

    def forward(self, x):                       # decoder.py:5-11
        return x.view(-1, *self.shape)


class _PlusMinusOne(nn.Module):
    def forward(self, x):                       # decoder.py:13-15
        return x * 2 - 1


class Decoder32(nn.Sequential):
    """Linear(128,2048) -> (128,4,4) -> 2x[ConvT k4s2p1, BN, ReLU] -> ConvT k4s2p1 -> Sigmoid -> x*2-1
    (decoder.py:17-33)."""

    def __init__(self):
        super().__init__(nn.Linear(128, 2048), _Reshape(128, 4, 4),
                         nn.ConvTranspose2d(128, 64, 4, 2, 1), nn.BatchNorm2d(64, affine=True), nn.ReLU(inplace=True),
                         nn.ConvTranspose2d(64, 32, 4, 2, 1), nn.BatchNorm2d(32, affine=True), nn.ReLU(inplace=True),
                         nn.ConvTranspose2d(32, 3, 4, 2, 1), nn.Sigmoid(), _PlusMinusOne())
