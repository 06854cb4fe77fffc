"""Oracle step choreography: DCGAN / SRGAN / CycleGAN + white-box wrapper.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Restates reference
models/{base,dcgan,srgan,cyclegan,wrappers,util}.py on plain torch CPU ops.
"""
from collections import OrderedDict
from itertools import chain

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import optim
from torch.nn import DataParallel

from . import nets as _nets
from . import bbox as _bbox
from .sign import SignLossModel


class Cfg(dict):
    """Minimal attr-dict standing in for configs.Config (configs/__init__.py:4-44)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in {**(d or {}), **kw}.items():
            self[k] = Cfg(v) if isinstance(v, dict) and not isinstance(v, Cfg) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, Cfg) else v) for k, v in self.items()}


CPU = [torch.device('cpu')]


class _CpuDataParallel(nn.Module):
    """What ``DataParallel(net, device_ids=[None])`` degenerates to on a CPU-only machine: a
    passthrough that owns ``.module`` (state_dict keys get the 'module.' prefix).  On a machine that
    HAS a GPU, torch's DataParallel would move the module to cuda:0 even for device=[cpu]; the oracle
    must stay on the CPU there, hence this explicit shell."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


def _wrap(net, device):
    # models/dcgan.py:13-17
    if device[0].type == 'cpu':
        return _CpuDataParallel(net)
    return DataParallel(net.to(device[0]), device_ids=[d.index for d in device])


class Model:
    """models/base.py:4-44."""

    def __init__(self):
        self._modules = OrderedDict()

    def state_dict(self):
        return OrderedDict((k, m.state_dict()) for k, m in self._modules.items())

    def load_state_dict(self, sd, strict=False):
        for k, m in self._modules.items():
            if strict:
                assert k in sd, f'Missing key: {k}'
            if k in sd:
                m.load_state_dict(sd[k])


class DCGAN(Model):
    """models/dcgan.py:7-78."""

    def __init__(self, config, device=CPU, networks=_nets):
        super().__init__()
        self.device = device
        self.G = _wrap(getattr(networks, config.G)(), device)
        self.D = _wrap(getattr(networks, config.D)(), device)
        self.G.train()
        self.D.train()
        make = getattr(optim, config.opt)
        kw = config.opt_param.to_dict()
        self.optG = make(self.G.parameters(), **kw)
        self.optD = make(self.D.parameters(), **kw)
        self._modules.update(G=self.G, D=self.D, optG=self.optG, optD=self.optD)

    def forward_d(self, data):                   # dcgan.py:42-48
        self.latent = data['latent']
        self.real_sample = data['real_sample']
        self.fake_sample = self.G(self.latent)
        self.real_logits = self.D(self.real_sample)
        self.fake_logits = self.D(self.fake_sample.detach())

    def compute_d_loss(self):                    # dcgan.py:31-35 hinge
        self.LossR = F.relu(1. - self.real_logits).mean()
        self.LossF = F.relu(1. + self.fake_logits).mean()
        self.LossD = self.LossR + self.LossF

    def forward_g(self, data):                   # dcgan.py:50-52
        self.generated = data['fake_sample']
        self.gen_logits = self.D(self.generated)

    def compute_g_loss(self):                    # dcgan.py:37-40
        self.LossA = -self.gen_logits.mean()
        self.LossG = self.LossA

    def get_metrics(self):                       # dcgan.py:54-61
        return {'D/Sum': self.LossD.item(), 'D/Real': self.LossR.item(),
                'D/Fake': self.LossF.item(), 'G/Sum': self.LossG.item(),
                'G/Adv': self.LossA.item()}

    def update_d(self, data):                    # dcgan.py:63-69
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.LossD.backward()
        self.optD.step()

    def update_g(self, data, update=True):       # dcgan.py:71-78
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.LossG.backward()
            self.optG.step()


class VAE(Model):
    """models/vae.py:9-74: one Adam over decoder G and encoder D; loss = KL + BCE, both sum / batch."""

    def __init__(self, config, device=CPU, networks=_nets):
        super().__init__()
        self.device = device
        self.G = _wrap(getattr(networks, config.G)(), device)
        self.D = _wrap(getattr(networks, config.D)(), device)
        self.G.train()
        self.D.train()
        make = getattr(optim, config.opt)
        self.optG = make(chain(self.G.parameters(), self.D.parameters()), **config.opt_param.to_dict())
        self._modules.update(G=self.G, D=self.D, opt=self.optG)

    def forward_d(self, data):                   # vae.py:50-55
        self.real_sample = data['real_sample']
        self.latent, (self.mean, self.logvar) = self.D(self.real_sample)
        self.fake_sample = self.G(self.latent)
        self.generated = self.fake_sample

    def compute_d_loss(self): pass

    def forward_g(self, data): pass

    def compute_g_loss(self):                    # vae.py:36-48
        n = self.mean.size(0)
        self.kl_loss = ((self.mean ** 2 + self.logvar.exp() - 1 - self.logvar) / 2).sum() / n
        self.reconstruct = F.binary_cross_entropy((self.fake_sample + 1.) / 2., (self.real_sample + 1.) / 2.,
                                                  reduction='sum') / n
        self.LossG = self.kl_loss + self.reconstruct

    def get_metrics(self):                       # vae.py:59-64 (tensors there; floats here like the other models)
        return {'G/KL': self.kl_loss.item(), 'G/R': self.reconstruct.item(), 'G/Sum': self.LossG.item()}

    def update_d(self, data):                    # vae.py:66-67
        self.forward_d(data)

    def update_g(self, data, update=True):       # vae.py:69-75
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.LossG.backward()
            self.optG.step()


class SRGAN(Model):
    """models/srgan.py:7-107."""

    def __init__(self, config, device=CPU, networks=_nets):
        super().__init__()
        self.device = device
        self.G = _wrap(getattr(networks, config.G)(), device)
        self.D = _wrap(getattr(networks, config.D)(), device)
        self.V = _wrap(getattr(networks, config.V)(), device)
        self.G.train()
        self.D.train()
        self.V.eval()
        make = getattr(optim, config.opt)
        kw = config.opt_param.to_dict()
        self.optG = make(self.G.parameters(), **kw)
        self.optD = make(self.D.parameters(), **kw)
        self._modules.update(G=self.G, D=self.D, optG=self.optG, optD=self.optD)

    def forward_g(self, data):                   # srgan.py:67-74
        self.low_res, self.high_res, self.pretrain = data['low_res'], data['high_res'], data['pretrain']
        self.super_res = self.G(self.low_res)
        if not self.pretrain:
            self.gen_logits = self.D(self.super_res)

    def compute_g_loss(self):                    # srgan.py:46-59
        if self.pretrain:
            self.LossG = F.mse_loss(self.super_res, self.high_res)
        else:
            self.LossA = F.binary_cross_entropy_with_logits(self.gen_logits, torch.ones_like(self.gen_logits))
            self.LossX = F.mse_loss(self.V(self.super_res), self.V(self.high_res).detach())
            self.LossG = self.LossX + 1e-3 * self.LossA

    def forward_d(self, data):                   # srgan.py:61-65
        self.high_res, self.super_res = data['high_res'], data['super_res']
        self.real_logits = self.D(self.high_res)
        self.fake_logits = self.D(self.super_res.detach())

    def compute_d_loss(self):                    # srgan.py:33-44
        self.LossR = F.binary_cross_entropy_with_logits(self.real_logits, torch.ones_like(self.real_logits))
        self.LossF = F.binary_cross_entropy_with_logits(self.fake_logits, torch.zeros_like(self.fake_logits))
        self.LossD = self.LossR + self.LossF

    def get_metrics(self):                       # srgan.py:76-90
        if self.pretrain:
            return {'G/MSE': self.LossG.item(), 'G/Sum': self.LossG.item()}
        return {'D/Sum': self.LossD.item(), 'D/Real': self.LossR.item(), 'D/Fake': self.LossF.item(),
                'G/Sum': self.LossG.item(), 'G/Adv': self.LossA.item(), 'G/Con': self.LossX.item()}

    def update_d(self, data):
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.LossD.backward()
        self.optD.step()

    def update_g(self, data, update=True):
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.LossG.backward()
            self.optG.step()


class ImagePool(nn.Module):
    """models/util.py:5-35 (history buffer of generated images)."""

    def __init__(self, pool_size):
        super().__init__()
        self.pool_size = pool_size
        if pool_size > 0:
            self.register_buffer('images', torch.tensor([]))
            self.register_buffer('counts', torch.zeros([]))

    def load_state_dict(self, *args, **kwargs):
        self.images = torch.empty_like(args[0]['images'])
        super().load_state_dict(*args, **kwargs)

    def __call__(self, images):
        if self.pool_size <= 0:
            return images.detach()
        if self.counts < self.pool_size:
            self.images = torch.cat([self.images.to(images.device), images.detach()], 0)[:self.pool_size]
            self.counts += images.size(0)
            return images.detach()
        images = images.detach()
        swap = torch.rand(images.size(0)) > 0.5
        index = torch.randperm(self.pool_size)[:images.size(0)]
        old = self.images[index[swap]].clone()
        self.images[index[swap]] = images[swap].detach()
        images[swap] = old
        return images.detach()


class CycleGAN(Model):
    """models/cyclegan.py:10-165."""

    def __init__(self, config, device=CPU, networks=_nets):
        super().__init__()
        self.device = device
        fg, fd = getattr(networks, config.G), getattr(networks, config.D)
        self.GA, self.GB = _wrap(fg(), device), _wrap(fg(), device)
        self.DA, self.DB = _wrap(fd(), device), _wrap(fd(), device)
        self.poolA, self.poolB = ImagePool(config.pool_size), ImagePool(config.pool_size)
        for n in (self.GA, self.GB, self.DA, self.DB):
            n.train()
        self.lambda_A, self.lambda_B, self.lambda_idt = config.lambda_A, config.lambda_B, config.lambda_idt
        make = getattr(optim, config.opt)
        kw = config.opt_param.to_dict()
        self.optG = make(chain(self.GA.parameters(), self.GB.parameters()), **kw)
        self.optD = make(chain(self.DA.parameters(), self.DB.parameters()), **kw)
        half = config.epoch // 2
        decay = lambda e: 1.0 - max(0, e - half) / half          # cyclegan.py:50-51
        self.schedulerG = optim.lr_scheduler.LambdaLR(self.optG, lr_lambda=decay)
        self.schedulerD = optim.lr_scheduler.LambdaLR(self.optD, lr_lambda=decay)
        self.MSE, self.L1 = nn.MSELoss(), nn.L1Loss()
        self._modules.update(GA=self.GA, GB=self.GB, DA=self.DA, DB=self.DB, optG=self.optG,
                             optD=self.optD, schG=self.schedulerG, schD=self.schedulerD,
                             poolA=self.poolA, poolB=self.poolB)

    def forward_g(self, data):                   # cyclegan.py:91-105
        self.real_A, self.real_B = data['real_A'], data['real_B']
        self.fake_B = self.GA(self.real_A)
        self.fake_A = self.GB(self.real_B)
        self.rec_A = self.GB(self.fake_B)
        self.rec_B = self.GA(self.fake_A)
        self.idt_A = self.GA(self.real_B)
        self.idt_B = self.GB(self.real_A)
        self.GA_logits = self.DA(self.fake_B)
        self.GB_logits = self.DB(self.fake_A)

    def compute_g_loss(self):                    # cyclegan.py:118-134
        self.LossGA = self.MSE(self.GA_logits, torch.ones_like(self.GA_logits))
        self.LossGB = self.MSE(self.GB_logits, torch.ones_like(self.GB_logits))
        self.LossCycA = self.L1(self.rec_A, self.real_A) * self.lambda_A
        self.LossCycB = self.L1(self.rec_B, self.real_B) * self.lambda_B
        self.LossG = self.LossGA + self.LossGB + self.LossCycA + self.LossCycB
        if self.lambda_idt > 0:
            self.LossIdtA = self.L1(self.idt_A, self.real_B) * self.lambda_B
            self.LossIdtB = self.L1(self.idt_B, self.real_A) * self.lambda_A
            self.LossG = self.LossG + self.lambda_idt * (self.LossIdtA + self.LossIdtB)
        else:
            self.LossIdtA = self.LossIdtB = torch.zeros([])

    def forward_d(self, data):                   # cyclegan.py:107-116
        self.real_A, self.real_B = data['real_A'], data['real_B']
        self.fake_A = self.poolA(data['fake_A'])
        self.fake_B = self.poolB(data['fake_B'])
        self.RA_logits = self.DB(self.real_A)
        self.FA_logits = self.DB(self.fake_A.detach())
        self.RB_logits = self.DA(self.real_B)
        self.FB_logits = self.DA(self.fake_B.detach())

    def compute_d_loss(self):                    # cyclegan.py:136-143
        self.LossDRA = self.MSE(self.RB_logits, torch.ones_like(self.RB_logits))
        self.LossDFA = self.MSE(self.FB_logits, torch.zeros_like(self.FB_logits))
        self.LossDA = (self.LossDRA + self.LossDFA) * 0.5
        self.LossDRB = self.MSE(self.RA_logits, torch.ones_like(self.RA_logits))
        self.LossDFB = self.MSE(self.FA_logits, torch.zeros_like(self.FA_logits))
        self.LossDB = (self.LossDRB + self.LossDFB) * 0.5

    def get_metrics(self):                       # cyclegan.py:73-89
        g = lambda t: t.item()
        return {'G/A': g(self.LossGA), 'G/B': g(self.LossGB), 'G/CycA': g(self.LossCycA),
                'G/CycB': g(self.LossCycB), 'G/IdtA': g(self.LossIdtA), 'G/IdtB': g(self.LossIdtB),
                'G/Sum': g(self.LossG), 'D/RealA': g(self.LossDRA), 'D/FakeA': g(self.LossDFA),
                'D/SumA': g(self.LossDA), 'D/RealB': g(self.LossDRB), 'D/FakeB': g(self.LossDFB),
                'D/SumB': g(self.LossDB), 'LR': self.optG.param_groups[0]['lr']}

    def update_lr(self):
        self.schedulerG.step()
        self.schedulerD.step()

    def update_g(self, data, update=True):
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.LossG.backward()
            self.optG.step()

    def update_d(self, data):                    # cyclegan.py:158-165
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.LossDA.backward()
        self.LossDB.backward()
        self.optD.step()


class WhiteBoxWrapper:
    """models/wrappers.py:76-125 + Wrapper base (models/base.py:46-79)."""

    def __init__(self, model, config):
        self.model = model
        self.config = config
        target = getattr(model, config.target)
        self.loss_model = SignLossModel(target, config).to(model.device[0])
        model._modules['sign'] = self.loss_model   # wrappers.py:87 (inner dict via __getattr__)

    def __getattr__(self, key):                   # base.py:52-58: unknown attrs -> None
        if key in self.__dict__:
            return self.__dict__[key]
        model = self.__dict__.get('model')
        if model is not None and hasattr(model, key):
            return getattr(model, key)
        return None

    def state_dict(self):
        return self.model.state_dict()

    def load_state_dict(self, sd, strict=False):
        return self.model.load_state_dict(sd, strict=strict)

    def update_d(self, data):
        self.model.update_d(data)

    def update_g(self, data, update=True):        # wrappers.py:115-125
        self.model.update_g(data, update=False)
        self.inhibit = data.get('inhibit_wbox', False)
        target = getattr(self.model, self.config.target)
        self.LossG = self.model.LossG
        self.LossS = torch.zeros_like(self.LossG) if self.inhibit else self.loss_model(target)
        if hasattr(self.model, 'LossW'):          # wrappers.py:98-103
            self.Lambda, self.LossW = self.model.Lambda, self.model.LossW
        else:
            self.Lambda, self.LossW = 0, torch.zeros_like(self.LossS)
        if update:
            self.model.optG.zero_grad()
            (self.LossG + self.Lambda * self.LossW + self.LossS).backward()
            self.model.optG.step()

    def get_metrics(self):                        # wrappers.py:108-113
        m = self.model.get_metrics()
        if not self.inhibit:
            m['P/SignLoss'] = self.LossS.item()
            m['G/Sum'] += self.LossS.item()
        return m


class _NoBNStats:
    """models/util.py:55-69."""

    def __init__(self, net):
        self.net, self.saved = net, {}

    def __enter__(self):
        for n, m in self.net.named_modules():
            if isinstance(m, nn.BatchNorm2d):
                self.saved[n] = m.track_running_stats
                m.track_running_stats = False

    def __exit__(self, *a):
        for n, m in self.net.named_modules():
            if n in self.saved:
                m.track_running_stats = self.saved[n]


class BlackBoxWrapper:
    """models/wrappers.py:7-74 + Wrapper base (models/base.py:46-79)."""

    def __init__(self, model, config, tools=_bbox):
        self.model, self.config = model, config
        norm = config.normalized
        self.fn_inp = _wrap(getattr(tools, config.fn_inp.type)(config.fn_inp, normalized=norm), model.device)
        self.fn_out = _wrap(getattr(tools, config.fn_out.type)(config.fn_out, normalized=norm), model.device)
        self.Lambda = config['lambda']
        self.loss_fn = getattr(tools, config.loss_fn)(normalized=norm)
        self._modules = model._modules
        self._modules['fn_inp'] = self.fn_inp
        self._modules['fn_out'] = self.fn_out

    def __getattr__(self, key):                  # base.py:60-67: unknown attributes resolve to None
        model = self.__dict__.get('model')
        return getattr(model, key, None) if model is not None else None

    def state_dict(self):
        return Model.state_dict(self)

    def load_state_dict(self, sd, strict=False):
        return Model.load_state_dict(self, sd, strict)

    def update_d(self, data):
        self.model.update_d(data)

    def forward_g(self, data):                   # wrappers.py:43-56
        self.inhibit = data.get('inhibit_bbox', False)
        if self.inhibit:
            return
        x = getattr(self.model, self.config.input_var)
        y = getattr(self.model, self.config.output_var)
        with torch.no_grad():
            self.xwm = self.fn_inp(x.detach())
            self.ywm = self.fn_out(y.detach())
        G = getattr(self.model, self.config.target)
        with _NoBNStats(G):
            self.Gxwm = G(self.xwm)

    def compute_g_loss(self):                    # wrappers.py:36-41
        self.LossG = self.model.LossG
        self.LossW = torch.zeros_like(self.LossG) if self.inhibit else self.loss_fn(self.Gxwm, self.ywm)

    def get_metrics(self):                       # wrappers.py:58-63
        m = self.model.get_metrics()
        if not self.inhibit:
            m[f'P/{self.config.loss_fn.upper()}'] = self.LossW.item()
            m['G/Sum'] += self.Lambda * self.LossW.item()
        return m

    def update_g(self, data, update=True):       # wrappers.py:65-74
        self.model.update_g(data, update=False)
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.model.optG.zero_grad()
            (self.LossG + self.Lambda * self.LossW).backward()
            self.model.optG.step()
