"""Step choreography with the reference's class / method / attribute names (models/base.py,
models/dcgan.py, models/wrappers.py), driving the HIP networks, fused loss kernels, the multi-tensor
Adam and (when torch.distributed is initialised) the RCCL gradient reducer.

One process drives ONE GPU: ``device`` keeps the reference's list-of-devices signature but only
``device[0]`` is used; multi-GPU runs start one process per GPU (see bench.py / INTEGRATION.md).
"""
from abc import ABC, abstractmethod
from collections import OrderedDict
from contextlib import contextmanager
from itertools import chain

import torch

from . import _lib as L
from . import networks, ops, optim, tools
from .parallel import GradReducer, Replica, broadcast_module
from .tools import PAIR_LOSS_MAX, loss_pair, loss_sum, loss_value

import os

# CycleGAN: passes of one network over inputs that are all known up front run as one pass over the concatenated batch
# (exact: its norm layers are per sample).  A/B switch.
_BATCH_PASSES = os.environ.get('IPRGAN_BATCH_PASSES', '1') != '0'
_PAIR_D = os.environ.get('IPRGAN_PAIR_D', '1') != '0'       # A/B switch for the paired discriminator pass
_PAIR_LOSS = os.environ.get('IPRGAN_PAIR_LOSS', '1') != '0'    # A/B switch: both hinge terms of a paired pass in one launch each way

__all__ = ['Model', 'Wrapper', 'DCGAN', 'SRGAN', 'CycleGAN', 'VAE', 'ImagePool', 'BlackBoxWrapper',
           'WhiteBoxWrapper', 'DisableBatchNormStats']


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


@contextmanager
def _no_param_grads(*nets):
    """Run a discriminator pass whose PARAMETER gradients nobody consumes (the generator update): the
    reference computes and discards them (its next ``optD.zero_grad()`` precedes the next D backward);
    skipping them leaves every result identical and saves the weight-gradient GEMMs."""
    params = [p for n in nets for p in n.parameters() if p.requires_grad]
    for p in params:
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p in params:
            p.requires_grad_(True)


_CONST = {}


def _const(value, like):
    """A cached device scalar (the seed of ``loss.backward`` and the zero of an inhibited / absent loss term): autograd and
    ``zeros_like`` would fill a fresh one per call - a kernel launch each, inside the training step."""
    key = (float(value), like.device, like.dtype)
    t = _CONST.get(key)
    if t is None:
        t = _CONST[key] = torch.full([], float(value), device=like.device, dtype=like.dtype)
    return t


def _backward(loss):
    """``loss.backward()`` with the cached seed."""
    loss.backward(_const(1.0, loss))


def _step(opt, reducer):
    reducer.reduce()
    reducer.wait()
    opt.grad_scale = reducer.scale          # 1/world: the buckets hold the SUM over ranks
    opt.step()


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
        flat = self.__dict__.get('_pair_logits')
        if flat is not None and _PAIR_LOSS and flat.numel() <= 2 * PAIR_LOSS_MAX:
            # the paired pass's [2B] logits: both hinge terms and their sum in one launch each way (bit-identical)
            self.LossR, self.LossF, self.LossD = loss_pair(L.LOSS_HINGE_REAL, L.LOSS_HINGE_FAKE, flat, flat.numel() // 2)
            return
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
        fake = self.fake_sample.detach()
        if _PAIR_D and hasattr(self.D.module, 'forward_pair') and fake.shape == self.real_sample.shape \
                and self.D.module.can_pair(fake):
            # D(real) and D(fake) as one pass of twice the batch (each half with its own spectral-norm sigma)
            flat = self.D.module.forward_pair_flat(self.real_sample, fake)
            n = fake.shape[0]
            self.real_logits, self.fake_logits, self._pair_logits = flat[:n], flat[n:], flat
        else:
            self._pair_logits = None
            self.real_logits = self.D(self.real_sample)
            self.fake_logits = self.D(fake)

    def forward_g(self, data):
        self.generated = data['fake_sample']
        with _no_param_grads(self.D):
            self.gen_logits = self.D(self.generated)

    def get_metrics(self):
        return _fetch({'D/Sum': self.LossD, 'D/Real': self.LossR, 'D/Fake': self.LossF,
                       'G/Sum': self.LossG, 'G/Adv': self.LossA})

    def update_d(self, data):
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.reduceD.arm()
        _backward(self.LossD)
        _step(self.optD, self.reduceD)

    def update_g(self, data, update=True):
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.reduceG.arm()
            _backward(self.LossG)
            _step(self.optG, self.reduceG)


class VAE(Model):
    """models/vae.py:9-74 - encoder ``D`` + decoder ``G`` trained by ONE optimizer on KL + BCE reconstruction
    (both summed over elements and divided by the batch size); ``update_d`` only runs the forward pass."""

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
        params = list(chain(self.G.parameters(), self.D.parameters()))
        self.optG = opt_fn(params, **config.opt_param.to_dict())
        self.reduceG = GradReducer(params)

        self._modules['G'] = self.G
        self._modules['D'] = self.D
        self._modules['opt'] = self.optG

    def compute_d_loss(self): pass

    def compute_g_loss(self):
        inv_n = 1.0 / self.mean.size(0)
        self.kl_loss = (loss_sum(L.LOSS_KL_MEAN, self.mean, None, inv_n)
                        + loss_sum(L.LOSS_KL_LOGVAR, self.logvar, None, inv_n))
        self.reconstruct = loss_sum(L.LOSS_BCE_PM1, self.fake_sample, self.real_sample, inv_n)
        self.LossG = self.kl_loss + self.reconstruct

    def forward_d(self, data):
        self.real_sample = data['real_sample'].to(self.device[0], non_blocking=True)
        self.latent, (self.mean, self.logvar) = self.D(self.real_sample)
        self.fake_sample = self.G(self.latent)
        self.generated = self.fake_sample

    def forward_g(self, data): pass

    def get_metrics(self):
        return _fetch({'G/KL': self.kl_loss, 'G/R': self.reconstruct, 'G/Sum': self.LossG})

    def update_d(self, data):
        self.forward_d(data)

    def update_g(self, data, update=True):
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.reduceG.arm()
            _backward(self.LossG)
            _step(self.optG, self.reduceG)


class SRGAN(Model):
    """models/srgan.py:7-107 - BCE-with-logits adversarial terms, VGG19-feature MSE content loss
    (inputs fed raw in [0,1], no ImageNet normalisation), pixel MSE in the pre-training phase."""

    def __init__(self, config, device=[torch.device('cpu'), ]):
        super().__init__()
        self.device = device
        dev = device[0]
        self.G = Replica(getattr(networks, config.G)(), dev)
        self.D = Replica(getattr(networks, config.D)(), dev)
        self.V = Replica(getattr(networks, config.V)(), dev)
        self.G.train()
        self.D.train()
        self.V.eval()
        for net in (self.G, self.D, self.V):
            broadcast_module(net)
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
        self.LossR = loss_value(L.LOSS_BCE_ONES, self.real_logits)
        self.LossF = loss_value(L.LOSS_BCE_ZEROS, self.fake_logits)
        self.LossD = self.LossR + self.LossF

    def compute_g_loss(self):
        if self.pretrain:
            self.LossG = loss_value(L.LOSS_MSE, self.super_res, self.high_res)
        else:
            self.LossA = loss_value(L.LOSS_BCE_ONES, self.gen_logits)
            sr_feat = self.V(self.super_res)
            with torch.no_grad():
                hr_feat = self.V(self.high_res)
            self.LossX = loss_value(L.LOSS_MSE, sr_feat, hr_feat)
            self.LossG = self.LossX + 1e-3 * self.LossA

    def forward_d(self, data):
        self.high_res = self._dev(data['high_res'])
        self.super_res = data['super_res']
        self.real_logits = self.D(self.high_res)
        self.fake_logits = self.D(self.super_res.detach())

    def forward_g(self, data):
        self.low_res = self._dev(data['low_res'])
        self.high_res = self._dev(data['high_res'])
        self.pretrain = data['pretrain']
        self.super_res = self.G(self.low_res)
        if not self.pretrain:
            with _no_param_grads(self.D):
                self.gen_logits = self.D(self.super_res)

    def get_metrics(self):
        if self.pretrain:
            v = _fetch({'G/MSE': self.LossG})
            return {'G/MSE': v['G/MSE'], 'G/Sum': v['G/MSE']}
        return _fetch({'D/Sum': self.LossD, 'D/Real': self.LossR, 'D/Fake': self.LossF,
                       'G/Sum': self.LossG, 'G/Adv': self.LossA, 'G/Con': self.LossX})

    def update_d(self, data):
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.reduceD.arm()
        _backward(self.LossD)
        _step(self.optD, self.reduceD)

    def update_g(self, data, update=True):
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.reduceG.arm()
            _backward(self.LossG)
            _step(self.optG, self.reduceG)


class ImagePool(torch.nn.Module):
    """models/util.py:5-35 - history of generated images; swap decisions use torch's CPU RNG exactly as
    the reference (torch.rand / torch.randperm), the buffers live on the model's device."""

    def __init__(self, pool_size):
        super().__init__()
        self.pool_size = pool_size
        if self.pool_size > 0:
            self.register_buffer('images', torch.tensor([]))
            self.register_buffer('counts', torch.zeros([]))

    def load_state_dict(self, *args, **kwargs):
        self.images = torch.empty_like(args[0]['images'])
        super().load_state_dict(*args, **kwargs)
        self._host_counts = None                    # re-read from the loaded buffer at the next call

    def __call__(self, images):
        if self.pool_size <= 0:
            return images.detach()
        # ``counts`` is a registered buffer (it lives on the model's device and is part of the checkpoint, as in the
        # reference); the branch below is taken on a HOST mirror of it, so that update_d does not drain the GPU queue
        # once per step to compare a device scalar (it cost one full synchronisation per CycleGAN step)
        if getattr(self, '_host_counts', None) is None:
            self._host_counts = float(self.counts)
        self._last_n = images.size(0)
        if self._host_counts < self.pool_size:
            self.images = torch.cat([self.images.to(images.device), images.detach()], dim=0)[:self.pool_size, ...]
            self.counts += images.size(0)
            self._host_counts += images.size(0)
            return images.detach()
        images = images.detach()
        if not images.is_cuda:              # (host tensors: the reference's own indexing form)
            prob = torch.rand(images.size(0)) > 0.5
            index = torch.randperm(self.pool_size)[:images.size(0)]
            pool_images = self.images[index[prob]].clone()
            self.images[index[prob]] = images[prob].detach()
            images[prob] = pool_images
            return images.detach()
        # The swap itself reads its decisions from device memory (iprgan_pool_swap), so a captured step can replay it with
        # new draws: eager calls draw here, a replay draws in graphs.GraphedStep's pre-replay hook (``draw``) - the same
        # two CPU-RNG calls in the same order either way.
        if not torch.cuda.is_current_stream_capturing():
            self.draw(images.size(0))
        if not self.images.is_contiguous():
            self.images = self.images.contiguous()
        if images.is_contiguous():
            ops.pool_swap(images, self.images, self._draws[0], self._draws[1])
            return images
        out = images.contiguous()           # (a permuted view: the swapped batch is returned, the view keeps its values)
        ops.pool_swap(out, self.images, self._draws[0], self._draws[1])
        return out

    def full(self):
        """True once every call takes the swap branch (fixed buffers: the step can be captured)."""
        if self.pool_size <= 0:
            return True
        if getattr(self, '_host_counts', None) is None:
            self._host_counts = float(self.counts)
        return self._host_counts >= self.pool_size

    def draw(self, n):
        """The swap decisions of one call, drawn from torch's CPU generator exactly as models/util.py:28-29 does, written to
        this pool's device table in stream order."""
        prob = torch.rand(n) > 0.5
        index = torch.randperm(self.pool_size)[:n]
        assert index.numel() == n, f'ImagePool: batch {n} exceeds pool_size {self.pool_size}'
        t = self.table(n)
        ops.write_ints(t[0], index.tolist())
        ops.write_ints(t[1], [int(p) for p in prob.tolist()])

    def table(self, n):
        """int32 [2, >= n] on the pool's device: row 0 = history slot per image, row 1 = swap / keep."""
        dev = self.images.device
        if getattr(self, '_draws', None) is None or self._draws.shape[1] < n or self._draws.device != dev:
            self._draws = torch.zeros(2, max(n, 8), dtype=torch.int32, device=dev)
        return self._draws


class CycleGAN(Model):
    """models/cyclegan.py:10-165 - LSGAN (MSE) adversarial terms, L1 cycle and identity terms,
    linearly decaying learning rate, image history pools."""

    def __init__(self, config, device=[torch.device('cpu'), ]):
        super().__init__()
        self.device = device
        dev = device[0]
        fn_g, fn_d = getattr(networks, config.G), getattr(networks, config.D)
        self.GA, self.GB = Replica(fn_g(), dev), Replica(fn_g(), dev)
        self.DA, self.DB = Replica(fn_d(), dev), Replica(fn_d(), dev)
        # batching two passes of a network into one is exact only without batch statistics
        per_sample = lambda net: not any(isinstance(m, torch.nn.modules.batchnorm._BatchNorm) for m in net.modules())
        self._batch_g = _BATCH_PASSES and per_sample(self.GA) and per_sample(self.GB)
        self._batch_d = _BATCH_PASSES and per_sample(self.DA) and per_sample(self.DB)
        self.poolA, self.poolB = ImagePool(config.pool_size), ImagePool(config.pool_size)
        for net in (self.GA, self.GB, self.DA, self.DB):
            net.train()
            broadcast_module(net)
        self.lambda_A, self.lambda_B, self.lambda_idt = config.lambda_A, config.lambda_B, config.lambda_idt

        opt_fn = getattr(optim, config.opt)
        opt_param = config.opt_param.to_dict()
        self.optG = opt_fn(chain(self.GA.parameters(), self.GB.parameters()), **opt_param)
        self.optD = opt_fn(chain(self.DA.parameters(), self.DB.parameters()), **opt_param)
        self.reduceG = GradReducer(list(chain(self.GA.parameters(), self.GB.parameters())))
        self.reduceD = GradReducer(list(chain(self.DA.parameters(), self.DB.parameters())))

        half_epoch = config.epoch // 2
        linear_lr = lambda e: 1.0 - max(0, e - half_epoch) / half_epoch
        self.schedulerG = optim.lr_scheduler.LambdaLR(self.optG, lr_lambda=linear_lr)
        self.schedulerD = optim.lr_scheduler.LambdaLR(self.optD, lr_lambda=linear_lr)

        for k, v in (('GA', self.GA), ('GB', self.GB), ('DA', self.DA), ('DB', self.DB),
                     ('optG', self.optG), ('optD', self.optD), ('schG', self.schedulerG),
                     ('schD', self.schedulerD), ('poolA', self.poolA), ('poolB', self.poolB)):
            self._modules[k] = v

    def _dev(self, t):
        return t.to(self.device[0], non_blocking=True)

    def get_metrics(self):
        m = _fetch({'G/A': self.LossGA, 'G/B': self.LossGB, 'G/CycA': self.LossCycA, 'G/CycB': self.LossCycB,
                    'G/IdtA': self.LossIdtA.to(self.device[0]), 'G/IdtB': self.LossIdtB.to(self.device[0]),
                    'G/Sum': self.LossG, 'D/RealA': self.LossDRA, 'D/FakeA': self.LossDFA, 'D/SumA': self.LossDA,
                    'D/RealB': self.LossDRB, 'D/FakeB': self.LossDFB, 'D/SumB': self.LossDB})
        m['LR'] = self.optG.param_groups[0]['lr']
        return m

    def forward_g(self, data):
        self.real_A = self._dev(data['real_A'])
        self.real_B = self._dev(data['real_B'])
        if self._batch_g:
            # GA(real_A) and GA(real_B) (the identity term, models/cyclegan.py:109-110) are two passes of the same network
            # whose inputs are both known here, and InstanceNorm statistics are per sample: ONE pass over the
            # concatenated batch computes the same values (launches of twice the size: the 256->256 layers have 1024
            # tiles instead of 512, four blocks per CU instead of two) and its backward the same weight gradients.
            n = self.real_A.shape[0]
            out = self.GA(torch.cat([self.real_A, self.real_B]))
            self.fake_B, self.idt_A = out[:n], out[n:]
            # ... and G_B's cycle pass over fake_B, known by now, joins G_B's other two (one pass of three batches)
            out = self.GB(torch.cat([self.real_B, self.real_A, self.fake_B]))
            self.fake_A, self.idt_B, self.rec_A = out[:n], out[n:2 * n], out[2 * n:]
            self.rec_B = self.GA(self.fake_A)
        else:
            self.fake_B = self.GA(self.real_A)
            self.fake_A = self.GB(self.real_B)
            self.rec_A = self.GB(self.fake_B)
            self.rec_B = self.GA(self.fake_A)
            self.idt_A = self.GA(self.real_B)
            self.idt_B = self.GB(self.real_A)
        with _no_param_grads(self.DA, self.DB):
            self.GA_logits = self.DA(self.fake_B)
            self.GB_logits = self.DB(self.fake_A)

    def forward_d(self, data):
        self.real_A = self._dev(data['real_A'])
        self.real_B = self._dev(data['real_B'])
        self.fake_A = self.poolA(data['fake_A'])
        self.fake_B = self.poolB(data['fake_B'])
        if self._batch_d:       # same argument for the two passes of each discriminator (InstanceNorm or no norm at all)
            n = self.real_A.shape[0]
            out = self.DB(torch.cat([self.real_A, self.fake_A.detach()]))
            self.RA_logits, self.FA_logits = out[:n], out[n:]
            out = self.DA(torch.cat([self.real_B, self.fake_B.detach()]))
            self.RB_logits, self.FB_logits = out[:n], out[n:]
        else:
            self.RA_logits = self.DB(self.real_A)
            self.FA_logits = self.DB(self.fake_A.detach())
            self.RB_logits = self.DA(self.real_B)
            self.FB_logits = self.DA(self.fake_B.detach())

    def compute_g_loss(self):
        self.LossGA = loss_value(L.LOSS_MSE_ONES, self.GA_logits)
        self.LossGB = loss_value(L.LOSS_MSE_ONES, self.GB_logits)
        self.LossCycA = loss_value(L.LOSS_L1, self.rec_A, self.real_A) * self.lambda_A
        self.LossCycB = loss_value(L.LOSS_L1, self.rec_B, self.real_B) * self.lambda_B
        self.LossG = self.LossGA + self.LossGB + self.LossCycA + self.LossCycB
        if self.lambda_idt > 0:
            self.LossIdtA = loss_value(L.LOSS_L1, self.idt_A, self.real_B) * self.lambda_B
            self.LossIdtB = loss_value(L.LOSS_L1, self.idt_B, self.real_A) * self.lambda_A
            self.LossG = self.LossG + self.lambda_idt * (self.LossIdtA + self.LossIdtB)
        else:
            self.LossIdtA = self.LossIdtB = torch.zeros([])

    def compute_d_loss(self):
        self.LossDRA = loss_value(L.LOSS_MSE_ONES, self.RB_logits)
        self.LossDFA = loss_value(L.LOSS_MSE_ZEROS, self.FB_logits)
        self.LossDA = (self.LossDRA + self.LossDFA) * 0.5
        self.LossDRB = loss_value(L.LOSS_MSE_ONES, self.RA_logits)
        self.LossDFB = loss_value(L.LOSS_MSE_ZEROS, self.FA_logits)
        self.LossDB = (self.LossDRB + self.LossDFB) * 0.5

    def update_lr(self):
        self.schedulerG.step()
        self.schedulerD.step()

    # ---- hooks of graphs.GraphedStep: what decides on the host in this model and how a captured step is fed
    def graph_ready(self):
        """Capturable once both history pools are full: while they fill, ``images`` is re-allocated by every call."""
        if not (self.poolA.full() and self.poolB.full()):
            return False
        for pool in (self.poolA, self.poolB):       # the tables exist before a capture could allocate them in its own pool
            if pool.pool_size > 0:
                pool.table(pool._last_n)
        return True

    def graph_before_replay(self):
        """The per-step host decisions of update_d (models/util.py:28-29), in the order forward_d consumes them."""
        for pool in (self.poolA, self.poolB):
            if pool.pool_size > 0:
                pool.draw(pool._last_n)

    def graph_signature(self):
        """What a captured step has baked in besides the optimizers' hyper-parameters: the pools' buffers (a loaded
        checkpoint replaces them) - a change makes GraphedStep capture again."""
        return tuple((p.images.data_ptr(), getattr(p, '_draws', None) is not None and p._draws.data_ptr())
                     for p in (self.poolA, self.poolB) if p.pool_size > 0)

    def update_g(self, data, update=True):
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.optG.zero_grad()
            self.reduceG.arm()
            _backward(self.LossG)
            _step(self.optG, self.reduceG)

    def update_d(self, data):
        self.forward_d(data)
        self.compute_d_loss()
        self.optD.zero_grad()
        self.reduceD.arm()
        _backward(self.LossDA)
        _backward(self.LossDB)
        _step(self.optD, self.reduceD)


class DisableBatchNormStats(object):
    """models/util.py:55-69 - inside the block every BatchNorm2d of ``model`` normalises with batch
    statistics and leaves its running statistics untouched (used by the black-box watermark pass)."""

    def __init__(self, model):
        self.model, self.cache = model, {}

    def __enter__(self):
        for name, m in self.model.named_modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                self.cache[name] = m.track_running_stats
                m.track_running_stats = False

    def __exit__(self, *args):
        for name, m in self.model.named_modules():
            if name in self.cache:
                m.track_running_stats = self.cache[name]


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


class BlackBoxWrapper(Wrapper):
    """models/wrappers.py:7-74 - trigger-set watermark: the target generator must map the transformed input
    ``fn_inp(x)`` to the watermarked output ``fn_out(G(x))``; adds ``lambda * loss_fn(G(fn_inp(x)), fn_out(y))``
    (a second generator forward/backward per step, batch statistics only) to the G objective."""

    def __init__(self, model, config):
        super().__init__(model, config)
        self.configure()

    def configure(self):
        normalized = self.config.normalized
        dev = self.device[0]
        make = lambda c: Replica(getattr(tools, c.type)(c, normalized=normalized), dev)
        self.fn_inp = make(self.config.fn_inp)
        self.fn_out = make(self.config.fn_out)
        self.Lambda = self.config['lambda']
        self.loss_fn = getattr(tools, self.config.loss_fn)(normalized=normalized)
        self._modules = self.model._modules
        self._modules['fn_inp'] = self.fn_inp
        self._modules['fn_out'] = self.fn_out

    def compute_g_loss(self):
        self.LossG = self.model.LossG
        self.LossW = _const(0.0, self.LossG) if self.inhibit else self.loss_fn(self.Gxwm, self.ywm)

    def forward_g(self, data):
        self.inhibit = data.get('inhibit_bbox', False)
        if self.inhibit:
            return
        x = getattr(self.model, self.config.input_var)
        y = getattr(self.model, self.config.output_var)
        with torch.no_grad():
            self.xwm = self.fn_inp(x.detach())
            self.ywm = self.fn_out(y.detach())
        G = getattr(self.model, self.config.target)
        with DisableBatchNormStats(G):
            self.Gxwm = G(self.xwm)

    def get_metrics(self):
        metrics = self.model.get_metrics()
        if not self.inhibit:
            w = self.LossW.item()
            metrics[f'P/{self.config.loss_fn.upper()}'] = w
            metrics['G/Sum'] += self.Lambda * w
        return metrics

    def update_g(self, data, update=True):
        self.model.update_g(data, update=False)
        self.forward_g(data)
        self.compute_g_loss()
        if update:
            self.model.optG.zero_grad()
            loss = self.LossG + self.Lambda * self.LossW
            red = self.model.reduceG
            red.arm()
            _backward(loss)
            _step(self.model.optG, red)


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
        self.LossS = _const(0.0, self.LossG) if self.inhibit else self.loss_model(target)
        if hasattr(self.model, 'LossW'):
            self.Lambda, self.LossW = self.model.Lambda, self.model.LossW
        else:
            self.Lambda, self.LossW = 0, _const(0.0, self.LossS)

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
            # (a term with weight 0 adds exactly nothing: skipped, not multiplied)
            loss = self.LossG + self.LossS if self.Lambda == 0 else self.LossG + self.Lambda * self.LossW + self.LossS
            red = self.model.reduceG
            red.arm()
            _backward(loss)
            _step(self.model.optG, red)
