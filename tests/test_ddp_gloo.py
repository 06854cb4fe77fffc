"""CPU, world_size 2 over gloo: the data-parallel plumbing of the step (flat gradient buckets owned by the reducer,
bucket views installed as ``.grad``, all-reduce(SUM) with 1/N folded into the optimizer, parameter broadcast at
start-up, last-pass detection and bucket send order) - the N>1 path of bench.py without GPUs."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    for p in (ROOT, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from iprgan.parallel import GradReducer, broadcast_module
    torch.manual_seed(100 + rank)                              # ranks start different on purpose
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    frozen = torch.nn.Parameter(torch.ones(4))                 # a parameter that gets no gradient
    broadcast_module(net)
    flat0 = torch.cat([t.flatten().float() for t in list(net.parameters()) + list(net.buffers())])
    params = list(net.parameters()) + [frozen]
    x = torch.randn(6, 5)                                      # each rank its own shard of the batch

    def backward():
        net(x).square().mean().backward()

    for p in params:
        p.grad = None
    backward()
    want = []
    for p in params:                                           # expected: mean over ranks of the local grads
        if p.grad is None:
            want.append(None)
            continue
        g = p.grad.clone()
        dist.all_reduce(g)
        want.append(g / world)

    def matches():
        ok = True
        for p, w in zip(params, want):
            if w is None:
                ok = ok and p.grad is None                     # nobody produced a gradient: stays None, Adam skips it
            else:
                ok = ok and torch.allclose(p.grad * red.scale, w, atol=1e-7)   # .grad holds the SUM, scale = 1/world
        return ok

    red = GradReducer(params, bucket_mb=0.0001)                # tiny buckets -> several buckets
    for p in params:
        p.grad = None                                          # opt.zero_grad()
    red.arm()                                                  # .grad = bucket views; autograd accumulates in place
    in_bucket = all(p.grad.data_ptr() == red.view_of(p).data_ptr() for p in params)
    backward()
    red.reduce()
    red.wait()
    ok = matches() and in_bucket and red.scale == 1.0 / world
    nb = len(red.buckets)
    for p in params:
        p.grad = None
    red.arm()                                                  # a second step reuses the buckets
    backward()
    red.reduce()
    red.wait()
    ok = ok and matches()
    gathered = [torch.zeros_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        torch.save({'ok': ok, 'same': same, 'buckets': nb}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_allreduce_and_broadcast_world2(tmp_path):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['ok'], 'summed gradients / scale wrong'
    assert res['same'], 'replicas differ after broadcast'
    assert res['buckets'] > 1


def test_world1_views_and_untouched_params():
    from iprgan.parallel import GradReducer
    p, q = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
    r = GradReducer([p, q])
    r.arm()
    assert p.grad.data_ptr() == r.view_of(p).data_ptr() and float(p.grad.abs().sum()) == 0.0
    (p * 2.0).sum().backward()                                 # autograd accumulates into the view in place
    r.reduce(); r.wait()
    assert torch.equal(p.grad, torch.full_like(p, 2.0)) and p.grad.data_ptr() == r.view_of(p).data_ptr()
    assert q.grad is None and r.scale == 1.0


def test_last_pass_detection_and_bucket_order():
    """The executor protocol on the host: two recorded passes, buckets leave only during the LAST pass, in the order
    their parameters are reported done (reverse layer order), and a late autograd contribution is refused."""
    from iprgan.parallel import GradReducer
    ps = [torch.nn.Parameter(torch.zeros(1000)) for _ in range(4)]         # "layers" 0..3
    r = GradReducer(ps, bucket_mb=4000 * 2 / (1 << 20))                     # two parameters per bucket
    r.trace = []
    r.note_forward(); r.note_forward()                                      # e.g. D(real), D(fake)
    r.arm()
    assert len(r.buckets) == 2 and r.buckets[0]['params'] == [ps[3], ps[2]]
    assert not r.begin_pass()                                               # first pass: nothing may leave
    for p in reversed(ps):
        r.view_of(p).add_(1.0); r.touch(p); r.params_done([p])
    r.end_pass()
    assert r.trace == []
    assert r.begin_pass()                                                   # second = last pass
    r.view_of(ps[3]).add_(1.0); r.params_done([ps[3]])
    assert r.trace == []
    assert r.bucket_would_complete([ps[2]])
    r.view_of(ps[2]).add_(1.0); r.params_done([ps[2]])
    assert [t[:2] for t in r.trace] == [('launch', 0)]                      # last layers' bucket left first
    r.view_of(ps[1]).add_(1.0); r.params_done([ps[1]])
    r.end_pass()                                                            # layer 0 not reported: flushed at the end
    assert [t[:2] for t in r.trace] == [('launch', 0), ('launch', 1)]
    try:
        (ps[0] * 1.0).sum().backward()
        raised = False
    except RuntimeError:
        raised = True
    assert raised, 'a gradient arriving after its bucket was sent must be refused'
    r.reduce(); r.wait()
    assert float(ps[3].grad.sum()) == 2000.0 and r.pending == 0
