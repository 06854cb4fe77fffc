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

    # ---- fast path: every parameter of a group has a gradient and all share one step count (every step of the three
    # GANs).  The pointer / size tables of parameters and moments are built once, the step count is ONE host tensor
    # that every state entry references, and a step costs a list comprehension over the gradients' pointers instead of
    # ~5 us of per-parameter bookkeeping (CPU-tensor add, int(), dict lookups: 1.4 ms for CycleGAN's 280 tensors).
    def _shared(self, group):
        import ctypes as C
        from . import _lib as L
        params = group['params']
        sh = self._fast.get(id(group))
        if sh is not None and sh['n'] == len(params) and self.state[params[0]].get('step') is sh['step_t'] \
                and self.state[params[-1]].get('step') is sh['step_t']:
            # the cached pointer tables must still describe THESE tensors: every parameter and both moments of every
            # parameter (a `p.data = ...` or a hand-edited state entry in the middle of the list would otherwise keep the
            # kernel updating a stale buffer, silently)
            st = self.state
            if sh['ptrs'] == [p.data_ptr() for p in params] and \
                    sh['msum'] == sum(st[p]['exp_avg'].data_ptr() + st[p]['exp_avg_sq'].data_ptr() for p in params):
                return sh
        states = [self._init_state(p) for p in params]
        steps = {float(st['step']) for st in states}
        if len(steps) != 1:
            return None
        step_t = torch.tensor(steps.pop(), dtype=torch.float32)
        for st in states:
            st['step'] = step_t
        sh = {'n': len(params), 'step_t': step_t, 'step': int(step_t),
              'ptab': L.ptr_table(params), 'mtab': L.ptr_table([st['exp_avg'] for st in states]),
              'vtab': L.ptr_table([st['exp_avg_sq'] for st in states]),
              'sizes': (C.c_longlong * len(params))(*[p.numel() for p in params]),
              'ptrs': [p.data_ptr() for p in params],
              'msum': sum(st['exp_avg'].data_ptr() + st['exp_avg_sq'].data_ptr() for st in states)}
        self._fast[id(group)] = sh
        return sh

    def replayed(self):
        """Host bookkeeping for one optimizer step that a captured graph executed on the device: the step count the
        state_dict reports, and the parameters' version counters (version-keyed caches, stale-graph check)."""
        for group in self.param_groups:
            sh = getattr(self, '_fast', {}).get(id(group))
            if sh is None or not sh.get('in_graph'):
                # not part of the captured step (an empty group, an optimizer the body never steps - whether or not the eager
                # warm-up stepped it): nothing moved on the device, nothing to book-keep.  graphs.GraphedStep._capture
                # refuses, ONCE, a capture whose body left a group unstepped that the warm-up had stepped - never a crash or a
                # drifting host count behind replays that have already advanced device state
                continue
            sh['step'] += 1
            sh['step_t'].fill_(sh['step'])
            torch.autograd.graph.increment_version(group['params'])

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if not hasattr(self, '_fast'):
            self._fast = {}
        for group in self.param_groups:
            params = group['params']
            grads = [p.grad for p in params]
            if params and all(g is not None and g.is_contiguous() for g in grads):
                sh = self._shared(group)
                if sh is not None:
                    beta1, beta2 = group['betas']
                    if getattr(self, 'device_step', False):
                        # step count on the device (graphs.py): the launch carries no host step number.  The counter is
                        # created from the host count the first time (outside any capture: warm-up steps run eagerly)
                        if 'step_dev' not in sh:
                            sh['step_dev'] = torch.tensor([sh['step']], dtype=torch.int32, device=params[0].device)
                            sh['coef'] = torch.zeros(2, dtype=torch.float32, device=params[0].device)
                        sh['step'] += 1
                        sh['step_t'].fill_(sh['step'])
                        if torch.cuda.is_current_stream_capturing():
                            # this group's update is part of the graph being captured: replayed() book-keeps for exactly the
                            # groups that carry this mark (an optimizer stepped in the warm-up but not by the captured body has
                            # a device counter too - it must not be advanced on the host behind every replay)
                            sh['in_graph'] = True
                        ops.adam_step_tables_dev(sh['ptab'], grads, sh['mtab'], sh['vtab'], sh['sizes'], sh['n'],
                                                 group['lr'], beta1, beta2, group['eps'], group['weight_decay'],
                                                 sh['step_dev'], sh['coef'], self.grad_scale)
                    else:
                        sh.pop('step_dev', None)
                        sh['step'] += 1
                        sh['step_t'].fill_(sh['step'])
                        ops.adam_step_tables(sh['ptab'], grads, sh['mtab'], sh['vtab'], sh['sizes'], sh['n'], group['lr'],
                                             beta1, beta2, group['eps'], group['weight_decay'], sh['step'], self.grad_scale)
                    torch.autograd.graph.increment_version(params)
                    continue
                self._fast.pop(id(group), None)          # parameters were replaced (.to(), load): rebuild next time
            if getattr(self, 'device_step', False) and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                # (graphs.GraphedStep turns this into "stay eager": a captured slow-path step would replay with the
                # bias corrections of the capture-time step number)
                raise RuntimeError('optim.Adam: this step cannot be captured in a graph (a parameter without a gradient, a '
                                   'non-contiguous gradient or differing step counts take the host-side path)')
            buckets = {}
            for p in group['params']:
                if p.grad is None:
                    continue
                st = self._init_state(p)
                # (a fresh tensor: the fast path shares one step tensor between the entries of a group)
                st['step'] = torch.tensor(float(st['step']) + 1.0, dtype=torch.float32)
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
