"""Capture of a whole training step (update_d + update_g) in ONE HIP graph.

The reference's loop (experiments/image_generation.py:86-101) issues two optimizer steps per iteration; here those are
~170 kernel launches (DCGAN-64) driven from Python through ctypes.  On the fp32 workloads the GPU is the bottleneck, but
with bf16 activations a DCGAN-64 step is 3.0 ms of device work behind 3.9 ms of host work.  ``GraphedStep`` runs the step
eagerly for a few warm-up calls (autotuning, operand caches, lazily built tables), then captures one call with
``torch.cuda.graph`` - every launch of the library goes to torch's current stream, which is the capture stream, and the
autograd engine's backward thread inherits it - and from then on replays the graph: one launch per training step.

Data-parallel runs (N > 1) capture the same way when the buckets travel through the C ABI's communicator
(``iprgan_allreduce_bucket`` on the reducer's side stream: forked off and joined back inside the capture); with
torch.distributed as the transport the step stays eager.  Capture at N > 1 is OPT-IN (``allow_ddp``: bench.py ``--graph on``,
train.py ``engine: {graph_ddp: true}``) until it has run on a real multi-GPU communicator: on this 1-GPU pool only a test
double (tests/stub_rccl.cpp) and the 1-rank real communicator have exercised it.  The outcome of a capture is AGREED across
ranks (all-reduce MIN of the success flag over torch.distributed): every rank replays or every rank stays eager.

What makes the step capturable:
  * inputs live in static device tensors (``copy_`` before each replay);
  * no host decision inside the step depends on device data; metrics are read after the step, from tensors the graph
    wrote (their Python bindings are restored after each replay, because an eager step in between rebinds them);
  * Adam's step number lives on the device (``optim.Adam.device_step``, ``iprgan_adam_step_dev``): a replay carries the
    kernel arguments of the captured call, so a host-computed bias correction would be frozen;
  * host bookkeeping that the skipped Python would have done is redone per replay: optimizer step counts for
    ``state_dict()``, parameter version counters (operand caches and the stale-graph check key on them).
  * host decisions that do NOT depend on device data are made before the replay and handed over in device memory:
    CycleGAN's ImagePool (models/util.py:27-34) draws its swap decisions from the CPU generator; the model exposes
    ``graph_ready()`` (pools full: fixed buffers), ``graph_before_replay()`` (draw + write the tables) and
    ``graph_signature()`` (buffers the graph has baked in); hyper-parameters that travel by value (a scheduler's learning
    rate) are compared before every replay and a change captures the step again.

Capturing does not execute: the captured call's device work happens at the first replay, its host side effects happened
during the capture - together they are exactly one step.  If the capture fails (an op that may not be captured), the
step stays eager and ``failed`` holds the reason: correctness never depends on the graph.
"""
import torch

from . import engine, optim


def _chain(model):
    """model, model.model, ... (the protection wrappers nest the GAN model)."""
    seen = []
    while model is not None and all(model is not s for s in seen):
        seen.append(model)
        nxt = model.__dict__.get('model', None)        # (an nn.Module attribute lives in _modules, not in __dict__)
        model = nxt if nxt is not None else getattr(model, '_modules', {}).get('model', None)
    return seen


def _optimizers(model):
    out = []
    for m in _chain(model):
        for v in list(m.__dict__.values()) + list(getattr(m, '_modules', {}).values()):
            if isinstance(v, optim.Adam) and all(v is not o for o in out):
                out.append(v)
    return out


class GraphedStep:
    """``step = GraphedStep(model, body, inputs)``; ``step(inputs)`` runs ``body(static_inputs)`` - eagerly ``warmup`` times,
    then as a captured graph.  ``body`` takes the dict of STATIC input tensors and performs one whole training step on
    ``model`` (e.g. ``update_d`` then ``update_g``).  ``step(inputs, eager=True)`` forces an eager step (profiling runs)."""

    def __init__(self, model, body, inputs, warmup=3, allow_ddp=False, capture_mode=None):
        self.model, self.body = model, body
        self.allow_ddp = bool(allow_ddp)
        # 'global' / 'thread_local' / None = by rank count (see _capture); IPRGAN_GRAPH_CAPTURE_MODE overrides None
        import os
        self.capture_mode = capture_mode or os.environ.get('IPRGAN_GRAPH_CAPTURE_MODE') or None
        self.static = {k: v.detach().clone() for k, v in inputs.items()}
        self.warm = int(warmup)
        self.graph, self.failed, self.replays = None, None, 0
        self.opts = _optimizers(model)
        if not self.opts:
            raise ValueError('GraphedStep: no iprgan.optim.Adam found on the model (its step count must move to the device)')
        for o in self.opts:
            o.device_step = True
        self._bound = []
        # models with host-side decisions inside the step (looked up on the class: the wrappers' __getattr__ delegates)
        self.hooks = [m for m in _chain(model) if callable(getattr(type(m), 'graph_before_replay', None))]

    def _snapshot(self):
        self._bound = [(m, k, v) for m in _chain(self.model) for k, v in m.__dict__.items() if isinstance(v, torch.Tensor)]

    def _capture(self):
        from . import parallel
        rank, nranks = parallel.world()
        # (both refusals below are decided from facts every rank shares - the flag, the collectively chosen transport - so
        # no rank enters the agreement collective further down alone)
        if nranks > 1 and not self.allow_ddp:
            self.failed = 'capture at N > 1 is opt-in (bench.py --graph on / engine.graph_ddp): unproven on a real multi-GPU communicator'
            return False
        if nranks > 1 and parallel.transport_name() != 'rccl-abi':
            # the gradient exchange is part of the step: through the C ABI's communicator it is forked / joined inside the
            # capture (parallel.GradReducer._launch); torch.distributed's collectives (gloo test path, fallback) are not
            self.failed = f'gradient exchange over {parallel.transport_name()} cannot be captured'
            return False
        torch.cuda.synchronize()
        if nranks > 1:
            parallel.sync_autotune()              # the tiles the graph bakes in are the same on every rank
        # version-keyed operand caches must not be HIT inside the capture (engine.drop_operand_caches)
        for m in _chain(self.model):
            for sub in (m.__dict__.get('_modules') or {}).values():
                if isinstance(sub, torch.nn.Module):
                    engine.drop_operand_caches(sub)
        g = torch.cuda.CUDAGraph()
        # host step counts as they stand: optimizers that already ran inside an aborted capture have counted a step the
        # device never executed (a checkpoint written later would resume with bias corrections one step ahead)
        saved = [(sh, sh['step']) for o in self.opts for sh in getattr(o, '_fast', {}).values()]
        for sh, _ in saved:
            sh.pop('in_graph', None)              # set again by Adam.step for every group the captured body steps
        try:
            # N > 1: RCCL's proxy threads make runtime calls of their own while this thread captures; in the default 'global'
            # mode any such call from ANY thread invalidates the capture ("operation not permitted when stream is capturing"),
            # 'thread_local' only polices the capturing thread (what PyTorch prescribes for NCCL inside graphs)
            mode = self.capture_mode or ('thread_local' if nranks > 1 else 'global')
            with torch.cuda.graph(g, capture_error_mode=mode):
                self.body(self.static)
        except Exception as e:                    # not capturable here: stay eager (the half-captured call did no device work)
            self.failed = f'{type(e).__name__}: {e}'
        if self.failed is None:
            # every optimizer the replay will book-keep for must have stepped on the device-counter path inside the capture:
            # a group the body never steps (or stepped on the host path) would make every replay advance device state and
            # then fail in Adam.replayed() on the host - decided HERE, once, not after the first replay
            # (`step_dev` alone does not say that: the eager warm-up creates it.  Adam.step marks a group `in_graph` when it
            # runs under capture; a group the warm-up stepped but the captured body did not would otherwise be book-kept as
            # stepped on every replay while the graph never touches it)
            missing = [i for i, o in enumerate(self.opts) for gr in o.param_groups
                       if gr['params'] and not getattr(o, '_fast', {}).get(id(gr), {}).get('in_graph')]
            if missing:
                self.failed = (f'optimizer(s) {sorted(set(missing))} have a parameter group that the captured step did not update on '
                               'the device-counter path (stepped only in the warm-up, on the host path, or not at all)')
        if nranks > 1:
            # one rank eager while its peers replay would still exchange matching buckets, but a rank that aborted mid-capture
            # and one that did not must not disagree about WHAT the next call does: agree on the outcome
            ok = parallel._all_ranks_ok(self.failed is None, self._agree_device())
            if not ok and self.failed is None:
                self.failed = 'a peer rank could not capture the step'
        if self.failed is not None:
            del g
            for sh, step in saved:
                sh['step'] = step
                sh['step_t'].fill_(step)
            self._reset_reducers()
            torch.cuda.synchronize()
            return False
        self.graph = g
        self._lrs = self._hyper()
        self._snapshot()
        return True

    def _agree_device(self):
        """Device of the flag tensor of the agreement collective: the GPU under torch.distributed's nccl backend, the host
        under gloo (test ranks sharing one GPU)."""
        import torch.distributed as dist
        return torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')

    def _reset_reducers(self):
        """After an aborted capture: the reducers' per-step state belongs to the dead capture (a pending-pass count that
        never counted down, bucket events recorded into the discarded graph).  The next eager step arms them afresh."""
        from . import parallel
        seen = {}
        for o in self.opts:
            for gr in o.param_groups:
                for p in gr['params']:
                    r = parallel.owner_of(p)
                    if r is not None:
                        seen[id(r)] = r
        for r in seen.values():
            r.armed, r.in_final, r.pending = False, False, 0
            for b in (r.buckets or ()):
                b['left'], b['launched'], b['done'] = 0, False, None
                b.pop('ready', None)

    def _hyper(self):
        """Hyper-parameters that travel BY VALUE in the captured Adam launches: a change (SRGAN's lr *= 0.1, a scheduler)
        must not be replayed over - the step is captured again."""
        return ([(g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']) for o in self.opts for g in o.param_groups]
                + [m.graph_signature() for m in self.hooks])

    def __call__(self, inputs=None, eager=False):
        if inputs is not None:
            for k, v in inputs.items():
                if v is not self.static[k]:
                    self.static[k].copy_(v, non_blocking=True)
        if eager or self.failed is not None or self.warm > 0 or not all(m.graph_ready() for m in self.hooks):
            if self.warm > 0 and not eager:
                self.warm -= 1
            return self.body(self.static)
        if self.graph is None:
            if not self._capture():
                return self.body(self.static)
            for m in self.hooks:
                m.graph_before_replay()
            self.graph.replay()                   # the captured call's device work (its host side ran during the capture)
            self.replays += 1
            return None
        if self._hyper() != self._lrs:            # learning rate, pool buffers (...) changed since the capture: capture afresh
            self.graph = None
            return self.__call__(None, eager=False)
        for m in self.hooks:
            m.graph_before_replay()
        self.graph.replay()
        self.replays += 1
        for o in self.opts:
            o.replayed()
        for m, k, v in self._bound:               # an eager step in between rebinds these to its own tensors
            m.__dict__[k] = v
        return None
