"""Step choreography with the reference's class / method / attribute names (models/base.py,
models/dcgan.py, models/wrappers.py), driving the HIP networks, fused loss kernels, the multi-tensor
Adam and (when torch.distributed is initialised) the RCCL gradient reducer.

One process drives ONE GPU: ``device`` keeps the reference's list-of-devices signature but only
``device[0]`` is used; multi-GPU runs start one process per GPU (see bench.py / INTEGRATION.md).
"""
from abc import ABC, abstractmethod
from collections import OrderedDict

import torch

from . import _lib as L
from . import networks, optim, tools
from .parallel import GradReducer, Replica, broadcast_module
from .tools import loss_value

__all__ = ['Model', 'Wrapper', 'DCGAN', 'WhiteBoxWrapper']


class Model(ABC):
    """models/base.py:4-44."""

    def __init__(self):
        self._modules = OrderedDict()

    def load_state_dict(self, state_dict, strict=False):
        for name, m in self._modules.items():
            if strict:
                assert name in state_dict, f'Missing key: {name}'
            if name in state_dict:
                m.load_state_dict(state_dict[name])

    def state_dict(self):
        return OrderedDict((name, m.state_dict()) for name, m in self._modules.items())

    @abstractmethod
    def update_d(self, data): pass

    @abstractmethod
    def update_g(self, data, update=True): pass

    @abstractmethod
    def get_metrics(self): pass

    def optimize_parameters(self, data):
        """Convenience only (not reference API): one D update followed by one G update."""
        self.update_d(data)
        self.update_g({'fake_sample': self.fake_sample})


def _fetch(scalars):
    """One packed device->host copy for a dict of 0-dim loss tensors (the reference pays one
    ``.item()`` sync per entry, models/dcgan.py:54-61)."""
    keys = list(scalars)
    vals = torch.stack([scalars[k].detach().reshape(()) for k in keys]).tolist()
    return dict(zip(keys, vals))


class DCGAN(Model):
    """models/dcgan.py:7-78 — hinge D loss, -mean G loss, Adam on both nets."""

    def __init__(self, config, device=[torch.device('cpu'), ]):
        super().__init__()
        self.device = device
        dev = device[0]
        self.G = Replica(getattr(networks, config.G)(), dev)
        self.D = Replica(getattr(networks, config.D)(), dev)
        self.G.train()
        self.D.train()
        broadcast_module(self.G)
        broadcast_module(self.D)

        opt_fn = getattr(optim, config.opt)
        opt_param = config.opt_param.to_dict()
        self.optG = opt_fn(self.G.parameters(), **opt_param)
        self.optD = opt_fn(self.D.parameters(), **opt_param)
        self.reduceG = GradReducer(list(self.G.parameters()))
        self.reduceD = GradReducer(list(self.D.parameters()))

        self._modules['G'] = self.G
        self._modules['D'] = self.D
        self._modules['optG'] = self.optG
        self._modules['optD'] = self.optD

    def _dev(self, t):
        return t.to(self.device[0], non_blocking=True)

    def compute_d_loss(self):
        self.LossR = loss_value(L.LOSS_HINGE_REAL, self.real_logits)
        self.LossF = loss_value(L.LOSS_HINGE_FAKE, self.fake_logits)
        self.LossD = self.LossR + self.LossF

    def compute_g_loss(self):
        self.LossA = loss_value(L.LOSS_NEG_MEAN, self.gen_logits)
        self.LossG = self.LossA

    def forward_d(self, data):
        self.latent = self._dev(data['latent'])
        self.real_sample = self._dev(data['real_sample'])
        self.fake_sample = self.G(self.latent)
        self.real_logits = self.D(self.real_sample)
        self.fake_logits = self.D(self.fake_sample.detach())

    def forward_g(self, data):
        self.generated = data['fake_sample']
        # D's parameter gradients from this pass are never consumed (optD.zero_grad() precedes the
        # next D backward), so only the data gradient is computed: identical results, fewer FLOPs.
        d_params = [p for p in self.D.parameters() if p.requires_grad]
        for p in d_params:
            p.requires_grad_(False)
        try:
            self.gen_logits = self.D(self.generated)
        finally:
            for p in d_params:
                p.requires_grad_(True)

    def get_metrics(self):
        return _fetch({'D/Sum': self.LossD, 'D/Real': self.LossR, 'D/Fake': self.LossF,
                       'G/Sum': self.LossG, 'G/Adv': self.LossA})

    def update_d(self, data):
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.LossD.backward()
        self.reduceD.reduce()
        self.reduceD.wait()
        self.optD.step()

    def update_g(self, data, update=True):
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.LossG.backward()
            self.reduceG.reduce()
            self.reduceG.wait()
            self.optG.step()


class Wrapper(Model):
    """models/base.py:46-79 — delegation; unknown attributes resolve to None, not AttributeError."""

    def __init__(self, model, config):
        self.model = model
        self.config = config

    def __getattr__(self, key):
        if key in self.__dict__:
            return self.__dict__[key]
        model = self.__dict__.get('model')
        if model is not None and hasattr(model, key):
            return getattr(model, key)
        return None

    def compute_d_loss(self):
        self.model.compute_d_loss()

    def forward_d(self, data):
        self.model.forward_d(data)

    def update_d(self, data):
        self.model.update_d(data)


class WhiteBoxWrapper(Wrapper):
    """models/wrappers.py:76-125 — adds the sign loss of the target generator to the G objective."""

    def __init__(self, model, config):
        super().__init__(model, config)
        self.configure()

    def configure(self):
        target = getattr(self.model, self.config.target)
        self.loss_model = tools.SignLossModel(target, self.config).to(self.device[0])
        self._modules['sign'] = self.loss_model

    def forward_g(self, data):
        self.inhibit = data.get('inhibit_wbox', False)

    def compute_g_loss(self):
        target = getattr(self.model, self.config.target)
        self.LossG = self.model.LossG
        self.LossS = torch.zeros_like(self.LossG) if self.inhibit else self.loss_model(target)
        if hasattr(self.model, 'LossW'):
            self.Lambda, self.LossW = self.model.Lambda, self.model.LossW
        else:
            self.Lambda, self.LossW = 0, torch.zeros_like(self.LossS)

    def get_metrics(self):
        metrics = self.model.get_metrics()
        if not self.inhibit:
            s = self.LossS.item()
            metrics['P/SignLoss'] = s
            metrics['G/Sum'] += s
        return metrics

    def update_g(self, data, update=True):
        self.model.update_g(data, update=False)
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.model.optG.zero_grad()
            loss = self.LossG + self.Lambda * self.LossW + self.LossS
            loss.backward()
            red = self.model.reduceG
            if red is not None:
                red.reduce()
                red.wait()
            self.model.optG.step()
