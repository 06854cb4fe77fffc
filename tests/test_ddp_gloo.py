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
    broadcast_module(net)
    flat0 = torch.cat([t.flatten().float() for t in list(net.parameters()) + list(net.buffers())])
    params = list(net.parameters())
    for i, p in enumerate(params):
        p.grad = torch.full_like(p, float((rank + 1) * (i + 1)))
    params[1].grad = None                                      # a parameter without gradient on this step
    red = GradReducer(params)
    red.reduce()
    red.wait()
    mean = sum(range(1, world + 1)) / world
    ok = all(torch.allclose(p.grad, torch.full_like(p, mean * (i + 1))) for i, p in enumerate(params) if i != 1)
    ok = ok and bool((params[1].grad == 0).all())
    gathered = [torch.zeros_like(flat0) for _ in range(world)]
    dist.all_gather(gathered, flat0)
    same = all(torch.equal(gathered[0], g) for g in gathered)
    if rank == 0:
        torch.save({'ok': ok, 'same': same}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_allreduce_and_broadcast_world2(tmp_path):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['ok'], 'averaged gradients wrong'
    assert res['same'], 'replicas differ after broadcast'


def test_world1_is_noop():
    from iprgan.parallel import GradReducer
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full_like(p, 2.0)
    r = GradReducer([p])
    r.reduce(); r.wait()
    assert torch.equal(p.grad, torch.full_like(p, 2.0))
