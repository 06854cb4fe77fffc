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
    """Averages the gradients of a parameter list across ranks (one exchange per
    optimizer step and network), overlapped with the backward pass.

    Parameters are packed, in REVERSE order (gradients become final from the last layer to the first),
    into flat fp32 buckets of ``bucket_mb``.  A post-accumulate-grad hook copies each finished gradient
    into its bucket; when a bucket is complete its ``all_reduce(SUM)`` is launched at once on a side
    stream (RCCL over xGMI with the nccl backend; gloo on CPU in the tests) while autograd keeps running
    the earlier layers' backward kernels on the compute stream.  ``wait()`` - called right before the
    optimizer step - joins the reductions, scales by 1/world and installs the averaged gradients as views
    of the buckets.  xGMI is point-to-point (7 links x ~153 GB/s): a few large buckets, not per-layer
    messages.  With world_size == 1 everything is a no-op and no hooks are installed.
    """

    def __init__(self, params, bucket_mb=16.0):
        self.params = [p for p in params if p.requires_grad]
        self.rank, self.world = world()
        self.buckets = None
        self.stream = None
        self._handles = []
        self._armed = False
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        if self.world > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- bucket layout -----------------------------------------------------------------------------
    def _ensure(self):
        if self.buckets is not None:
            return
        dev = self.params[0].device
        self.buckets, self.where = [], {}
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            if cur and cur_bytes + p.numel() * 4 > self.bucket_bytes:
                self._close(cur, dev)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += p.numel() * 4
        if cur:
            self._close(cur, dev)
        if dev.type == 'cuda':
            self.stream = torch.cuda.Stream(device=dev)

    def _close(self, plist, dev):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=torch.float32, device=dev)
        off, views = 0, []
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            self.where[p] = (len(self.buckets), len(views) - 1)
            off += p.numel()
        self.buckets.append({'flat': flat, 'params': plist, 'views': views, 'pending': len(plist), 'work': None})

    # -- per-step protocol ---------------------------------------------------------------------------
    def arm(self):
        """Call before backward: gradients produced from now on are reduced as they complete."""
        if self.world == 1:
            return
        self._ensure()
        for b in self.buckets:
            b['pending'], b['work'] = len(b['params']), None
        self._armed = True

    def _on_grad(self, p):
        if not self._armed:
            return
        bi, vi = self.where[p]
        b = self.buckets[bi]
        b['views'][vi].copy_(p.grad)
        b['pending'] -= 1
        if b['pending'] == 0:
            self._launch(b)

    def _launch(self, b):
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, async_op=True)
        else:
            b['work'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, async_op=True)

    def reduce(self):
        """Flush: buckets whose parameters did not all receive a gradient this step (frozen or unused
        parameters count as zero) are reduced now."""
        if self.world == 1 or not self._armed:
            return
        for b in self.buckets:
            if b['work'] is None:
                for p, v in zip(b['params'], b['views']):
                    if p.grad is None:
                        v.zero_()
                    elif b['pending'] > 0 and p.grad.data_ptr() != v.data_ptr():
                        v.copy_(p.grad)
                self._launch(b)

    def wait(self):
        """Join the reductions on the compute stream and install the averaged gradients."""
        if self.world == 1 or not self._armed:
            return
        self._armed = False
        for b in self.buckets:
            b['work'].wait()
            if self.stream is not None:
                torch.cuda.current_stream().wait_stream(self.stream)
            b['flat'].mul_(1.0 / self.world)
            for p, v in zip(b['params'], b['views']):
                p.grad = v
            b['work'] = None

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []


def broadcast_module(module, src=0):
    """Make every rank start from rank-``src``'s parameters and buffers (replicas live for the run)."""
    _, w = world()
    if w == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src)
