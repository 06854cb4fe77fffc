"""GPU: the data-parallel step of the real engine with CUDA tensors - two ranks share the one GPU of the test
box and exchange gradients over gloo (RCCL needs one GPU per rank; the 8-GPU run is the driver's).  Exercises
what the CPU gloo test cannot: post-accumulate-grad hooks fired from the HIP chain's backward, bucket copies and the
side-stream ordering against the compute stream, Adam on the averaged bucket views."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


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
    from iprgan import Config, models
    from oracle import cases, recipe
    dev = torch.device('cuda:0')
    torch.manual_seed(10 + rank)                               # different initial weights: broadcast must fix that
    m = models.WhiteBoxWrapper(models.DCGAN(Config(cases.DCGAN_CFG), device=[dev]), Config(cases.WBOX_CFG))
    for s in range(2):
        x = torch.tanh(recipe.tensor(50 + rank, s, (4, 3, 64, 64)))          # every rank its own shard
        z = recipe.tensor(60 + rank, s, (4, 128))
        m.update_d({'real_sample': x, 'latent': z})
        m.update_g({'fake_sample': m.fake_sample})
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().flatten() for n in (m.G, m.D) for p in n.parameters()]).cpu()
    gathered = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(gathered, flat)
    bufs = torch.cat([b.detach().flatten().float() for b in m.G.buffers()]).cpu()       # BN running stats stay local
    gb = [torch.zeros_like(bufs) for _ in range(world)]
    dist.all_gather(gb, bufs)
    if rank == 0:
        torch.save({'same_params': all(torch.equal(gathered[0], g) for g in gathered),
                    'local_bn': not torch.equal(gb[0], gb[1]),
                    'finite': bool(torch.isfinite(flat).all()),
                    'ber': float(m.loss_model.compute_ber(m.G)),
                    'buckets': (len(m.reduceG.buckets), len(m.reduceD.buckets))}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_dcgan_steps_two_ranks_on_one_gpu(tmp_path):
    out = str(tmp_path / 'res.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    res = torch.load(out)
    assert res['finite'] and res['ber'] == 0.0
    assert res['same_params'], 'replicas diverged: gradients were not averaged identically on both ranks'
    assert res['local_bn'], 'BatchNorm statistics are per-rank (the reference DataParallel does not sync them)'
    assert res['buckets'][0] >= 2 and res['buckets'][1] >= 1
