"""Optimisers looked up by name from the config (``getattr(optim, config.opt)``, models/dcgan.py:21).

``Adam`` keeps torch.optim.Adam's hyper-parameters, ``param_groups`` and ``state_dict`` layout
(state[p] = {'step', 'exp_avg', 'exp_avg_sq'}) so checkpoints interchange with the reference, but
``step()`` is one multi-tensor HIP launch (csrc/elementwise.hip: adam_kernel) over all parameters
instead of ATen's per-tensor foreach kernels.
"""
import torch
from torch.optim import Optimizer
from torch.optim import lr_scheduler  # noqa: F401  (models use optim.lr_scheduler.LambdaLR)

from . import ops


class Adam(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError('invalid Adam hyper-parameters')
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))
        # data-parallel runs hand over the SUM of the ranks' gradients; 1/world is applied inside the kernel
        # (parallel.GradReducer.scale), 1.0 otherwise
        self.grad_scale = 1.0

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st['step'] = torch.tensor(0.0, dtype=torch.float32)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def load_state_dict(self, state_dict):
        """torch's loader moves every state tensor to the parameter's device; ``step`` is host bookkeeping (the
        bias corrections are formed on the host) and goes back to the CPU, so that a resumed run does not pay a
        device round trip per parameter and step."""
        super().load_state_dict(state_dict)
        for st in self.state.values():
            if 'step' in st:
                v = st['step']
                st['step'] = torch.tensor(float(v), dtype=torch.float32)       # int (torch 1.8) or any-device tensor

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            buckets = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self._init_state(p)
                if not torch.is_tensor(st['step']) or st['step'].is_cuda:
                    st['step'] = torch.tensor(float(st['step']), dtype=torch.float32)
                st['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                buckets.setdefault(int(st['step']), []).append((p, g, st['exp_avg'], st['exp_avg_sq']))
            beta1, beta2 = group['betas']
            for step, items in buckets.items():
                ps, gs, ms, vs = zip(*items)
                ops.adam_step(list(ps), list(gs), list(ms), list(vs), group['lr'], beta1, beta2,
                              group['eps'], group['weight_decay'], step, self.grad_scale)
                # the kernel wrote through raw pointers: tell autograd the parameters changed (graphs recorded
                # before this step must not be back-propagated with the new weights; version-keyed caches refresh)
                torch.autograd.graph.increment_version(ps)
        return loss
