"""Data-parallel plumbing.

``Replica`` stands where the reference puts ``torch.nn.DataParallel`` (models/dcgan.py:16-17): it owns
``.module``, so state_dict keys keep the ``module.`` prefix and ``named_modules()`` keeps the names the
sign-loss buffers are derived from (tools/sign_model.py:36).  Unlike DataParallel it never replicates
or scatters: this engine runs ONE process per GPU, each with a full replica that lives for the whole
run, and averages gradients with RCCL all-reduce over xGMI (``GradReducer``).
"""
import torch
import torch.distributed as dist
import torch.nn as nn


class Replica(nn.Module):
    def __init__(self, module, device):
        super().__init__()
        self.device = torch.device(device)
        self.module = module.to(self.device)

    def forward(self, *inputs):
        moved = [t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in inputs]
        return self.module(*moved)


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class GradReducer:
    """Averages the gradients of a parameter list across ranks through ONE flat fp32 bucket
    (one exchange per optimizer step, D bucket then G bucket).

    The bucket is reduced with ``all_reduce(SUM)`` on a side stream (RCCL over xGMI when the backend
    is nccl; gloo on CPU for the tests) and scaled by 1/world; ``wait()`` makes the compute stream
    wait for it before Adam reads the gradients.  With world_size == 1 it is a no-op.
    """

    def __init__(self, params):
        self.params = [p for p in params]
        self.rank, self.world = world()
        self.flat = None
        self.stream = None
        self.work = None

    def _ensure(self):
        if self.flat is None:
            n = sum(p.numel() for p in self.params)
            dev = self.params[0].device
            self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
            if dev.type == 'cuda':
                self.stream = torch.cuda.Stream(device=dev)

    def reduce(self):
        """Launch the all-reduce of the current .grad tensors (call right after backward)."""
        if self.world == 1:
            return
        self._ensure()
        grads = [p.grad for p in self.params]
        off = 0
        views = []
        for p, g in zip(self.params, grads):
            v = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            views.append(v)
            if g is None:
                v.zero_()
            else:
                v.copy_(g)
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self.work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        self.views = views

    def wait(self):
        """Block the compute stream on the reduction and install the averaged gradients."""
        if self.world == 1 or self.work is None:
            return
        self.work.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        self.flat.mul_(1.0 / self.world)
        for p, v in zip(self.params, self.views):
            p.grad = v
        self.work = None


def broadcast_module(module, src=0):
    """Make every rank start from rank-``src``'s parameters and buffers (replicas live for the run)."""
    _, w = world()
    if w == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
