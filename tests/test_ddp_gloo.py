"""CPU, world_size 2 over gloo: the data-parallel plumbing of the step (flat gradient bucket,
all-reduce(SUM) x 1/N, parameter broadcast at start-up) - the N>1 path of bench.py without GPUs."""
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
        for p in params:
            p.grad = None
        net(x).square().mean().backward()

    backward()
    want = []
    for p in params:                                           # expected: mean over ranks of the local grads
        g = torch.zeros_like(p) if p.grad is None else p.grad.clone()
        dist.all_reduce(g)
        want.append(g / world)
    red = GradReducer(params, bucket_mb=0.0001)                # tiny buckets -> several buckets, hook path
    red.arm()
    backward()                                                 # hooks launch the bucket reductions
    red.reduce()
    red.wait()
    ok = all(torch.allclose(p.grad, w, atol=1e-7) for p, w in zip(params, want))
    nb = len(red.buckets)
    red.arm()                                                  # a second step reuses the buckets
    backward()
    red.reduce()
    red.wait()
    ok = ok and all(torch.allclose(p.grad, w, atol=1e-7) for p, w in zip(params, want))
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
    assert res['ok'], 'averaged gradients wrong'
    assert res['same'], 'replicas differ after broadcast'
    assert res['buckets'] > 1


def test_world1_is_noop():
    from iprgan.parallel import GradReducer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full_like(p, 2.0)
    r = GradReducer([p])
    r.arm(); r.reduce(); r.wait()
    assert torch.equal(p.grad, torch.full_like(p, 2.0))
