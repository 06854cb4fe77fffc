"""Thin training driver with the reference's command line and config files:

    python ipr-gan_amd/train.py -c <reference YAML>            (reference: train.py:10-49, README.md:36)
    python -m torch.distributed.run --nproc-per-node 8 ipr-gan_amd/train.py -c <yaml>     (one process per GPU)

It reproduces ``Experiment.start/train/checkpoint`` (experiments/base.py:70-82 and the three ``train()``
bodies, image_generation.py:86-101, image_super_resolution.py:84-113, image_translation.py:90-112) on the
HIP engine, writes ``checkpoint.pt`` in the reference layout (resumable both ways) and a metrics JSONL.
Out of scope: real datasets (no data on the box: synthetic batches of the configured
shapes instead), TensorBoard, FID/IS/PSNR evaluation.
"""
import argparse
import json
import math
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from iprgan import Config, configs, models  # noqa: E402


class SyntheticLoader:
    """Infinite ``next()``-able loader (datasets/util.py:3-15) producing batches of the configured shapes and
    value ranges: DCGAN images in [-1,1] (datasets/img_datasets.py:12-17), SRGAN (lr, hr) in [0,1]
    (datasets/sr_datasets.py:27-33), CycleGAN unaligned pairs in [-1,1]."""

    def __init__(self, kind, bsz, size, n_samples=10000):
        self.kind, self.bsz, self.size, self.n = kind, bsz, size, n_samples

    def __len__(self):
        return self.n

    def __next__(self):
        b, s = self.bsz, self.size
        if self.kind == 'generation':
            return torch.tanh(torch.randn(b, 3, s, s)), torch.zeros(b, dtype=torch.long)
        if self.kind == 'super_resolution':
            return torch.rand(b, 3, 24, 24), torch.rand(b, 3, 96, 96)
        return torch.tanh(torch.randn(b, 3, s, s)), torch.tanh(torch.randn(b, 3, s, s))


class Experiment:
    KINDS = {'ImageGeneration': 'generation', 'ImageSuperResolution': 'super_resolution',
             'ImageTranslation': 'translation'}

    def __init__(self, config):
        self.config = config
        self.kind = self.KINDS[config.experiment]
        self.rank = int(os.environ.get('RANK', 0))
        self.world = int(os.environ.get('WORLD_SIZE', 1))
        os.makedirs(config.log.path, exist_ok=True)
        if self.rank == 0:
            with open(os.path.join(config.log.path, 'config.yaml'), 'w') as f:
                f.write(config.to_yaml())                      # dumped BEFORE the mutations below (base.py:15-19)
        self.init_step = 1
        self.engine = configs.apply_engine(config)               # optional `engine:` block (math mode, buckets, switches)
        self.configure_device()
        self.configure_dataset()
        self.configure_model()
        self.configure_protection()

    def configure_device(self):
        """base.py:24-43, one process per GPU: the per-process batch stays hparam.bsz (the reference's
        DataParallel scatters bsz*ngpu over ngpu devices), iteration counts are divided by the world size."""
        if not torch.cuda.is_available():
            raise SystemExit('the HIP engine needs a GPU')
        local = int(os.environ.get('LOCAL_RANK', 0))
        if local >= torch.cuda.device_count() and os.environ.get('IPRGAN_SHARE_DEVICE') == '1':
            local %= torch.cuda.device_count()      # test-only: several ranks on one GPU (needs IPRGAN_DIST_BACKEND=gloo)
        torch.cuda.set_device(local)
        self.device = [torch.device('cuda', local)]
        if self.world > 1 and not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            backend = os.environ.get('IPRGAN_DIST_BACKEND', 'nccl')       # nccl = RCCL over xGMI
            if backend == 'nccl':
                dist.init_process_group('nccl', device_id=self.device[0])
            else:
                dist.init_process_group(backend)
        hp = self.config.hparam
        if 'pretrain_iter' in hp.to_dict():
            hp.pretrain_iter //= self.world
        hp.iteration //= self.world

    def configure_dataset(self):
        ds, hp = self.config.dataset, self.config.hparam
        size = ds.get('size', None) or ds.get('crop', None) or 96
        # (`dataset.synthetic_samples`: length of the synthetic stand-in dataset - the reference takes it from the files on
        # disk; it sets the iterations per epoch of the translation experiment)
        self.data_loader = SyntheticLoader(self.kind, hp.bsz, size, n_samples=int(ds.get('synthetic_samples', None) or 10000))
        if self.kind == 'translation':                          # iteration / log.freq are given in epochs
            # image_translation.py:38-40 with the reference's global batch bsz * ngpu (base.py:39): one epoch is
            # ceil(N / (bsz * world)) iterations of every rank
            n = math.ceil(len(self.data_loader) / (hp.bsz * self.world))
            hp.iteration *= n
            self.config.log.freq *= n

    def configure_model(self):
        mc = self.config.model
        if self.kind == 'translation':
            mc.epoch = self.config.hparam.iteration // self.config.log.freq     # image_translation.py:44
        self.model = getattr(models, mc.type)(mc, device=self.device)

    def configure_protection(self):
        wm = self.config.get('protection', None)
        self.wbox = False
        if not wm:
            return
        bbox = wm.get('bbox', None)
        if bbox:                                                # image_generation.py:60-68, image_super_resolution.py:59-66,
            bbox['normalized'], bbox['input_var'], bbox['output_var'], bbox['target'] = {      # image_translation.py:65-72
                'generation': (True, 'latent', 'generated', 'G'),
                'super_resolution': (False, 'low_res', 'super_res', 'G'),
                'translation': (True, 'real_B', 'fake_A', 'GB')}[self.kind]
            self.model = models.BlackBoxWrapper(self.model, bbox)
        wbox = wm.get('wbox', None)
        if wbox:
            wbox['target'] = 'GB' if self.kind == 'translation' else 'G'         # image_translation.py:83
            self.model = models.WhiteBoxWrapper(self.model, wbox)
            self.wbox = True

    # ---- one iteration: the three reference train() bodies -------------------------------------------------
    def train(self):
        hp, m = self.config.hparam, self.model
        d_iter, g_iter = hp.get('d_iter', 1), hp.get('g_iter', 1)
        if self.kind == 'generation' and self.engine.get('graph') and d_iter == 1 and g_iter == 1 \
                and not self.config.get('protection', Config({})).get('bbox', None):
            # `engine: {graph: true}`: update_d + update_g as ONE captured HIP graph (iprgan/graphs.py).  The data and the
            # latent draw stay on the host, as in the reference loop; they are copied into the graph's static inputs.
            x, _ = next(self.data_loader)
            z = torch.randn(x.size(0), 128)
            if getattr(self, '_graphed', None) is None:
                from iprgan import graphs
                dev = self.device[0]

                def body(s):
                    m.update_d({'real_sample': s['x'], 'latent': s['z']})
                    m.update_g({'fake_sample': m.fake_sample})
                self._graphed = graphs.GraphedStep(m, body, {'x': x.to(dev), 'z': z.to(dev)}, warmup=3,
                                                   allow_ddp=bool(self.engine.get('graph_ddp')))
            self._graphed({'x': x, 'z': z})
        elif self.kind == 'generation':
            for _ in range(d_iter):
                x, _ = next(self.data_loader)
                m.update_d({'real_sample': x, 'latent': torch.randn(x.size(0), 128)})
            for _ in range(g_iter):
                m.update_g({'fake_sample': m.fake_sample})
        elif self.kind == 'super_resolution':
            pre = hp.get('pretrain_iter', 0)
            if self._step == pre + hp.iteration // 2 and pre > 0:
                m.optG.param_groups[0]['lr'] *= 0.1
                m.optD.param_groups[0]['lr'] *= 0.1
            if self._step <= pre:
                lr, hr = next(self.data_loader)
                m.update_g({'low_res': lr, 'high_res': hr, 'pretrain': True, 'inhibit_bbox': True})
            else:
                for _ in range(g_iter):
                    lr, hr = next(self.data_loader)
                    m.update_g({'low_res': lr, 'high_res': hr, 'pretrain': False})
                for _ in range(d_iter):
                    m.update_d({'high_res': m.high_res, 'super_res': m.super_res})
        else:
            if self._step % self.config.log.freq == 1 and self._step > 1:
                m.update_lr()
            for _ in range(g_iter):
                a, b = next(self.data_loader)
                m.update_g({'real_A': a, 'real_B': b})
            for _ in range(d_iter):
                m.update_d({'real_A': m.real_A, 'real_B': m.real_B,
                            'fake_A': m.fake_A.detach(), 'fake_B': m.fake_B.detach()})

    def checkpoint(self, metrics_every=1):
        path = os.path.join(self.config.log.path, 'checkpoint.pt')
        if self._step == 'end':
            if self.rank == 0:
                sd = self.model.state_dict()
                sd['step'] = 'END'
                torch.save(sd, path)
            return
        if self._step % metrics_every == 0:
            metrics = self.model.get_metrics()
            if self.rank == 0:
                with open(os.path.join(self.config.log.path, 'metrics.jsonl'), 'a') as f:
                    f.write(json.dumps({'step': self._step, **metrics}) + '\n')
        if self._step % self.config.log.freq == 0 and self.rank == 0:
            sd = self.model.state_dict()
            sd['step'] = self._step
            torch.save(sd, path)

    def load_state_dict(self, state_dict, strict=False):
        self.model.load_state_dict(state_dict, strict=strict)
        if state_dict['step'] == 'END':
            self.init_step = self.config.hparam.get('pretrain_iter', 0) + self.config.hparam.iteration
        else:
            self.init_step = state_dict['step'] + 1

    def start(self, max_steps=None, metrics_every=1):
        total = self.config.hparam.get('pretrain_iter', 0) + self.config.hparam.iteration
        last = total if max_steps is None else min(total, self.init_step + max_steps - 1)
        for step in range(self.init_step, last + 1):
            self._step = step
            self.train()
            if step == self.init_step:
                from iprgan import parallel
                parallel.sync_autotune()          # N > 1: every rank adopts rank 0's tile choices (no-op on one rank)
            if step == self.init_step + 1:
                # long-lived objects out of the cyclic collector's sight: a generation-2 pass over torch's ~1 M objects
                # stalls the enqueueing thread for ~100 ms every few dozen iterations otherwise
                import gc
                gc.collect()
                gc.freeze()
            self.checkpoint(metrics_every)
        if last >= total:
            self._step = 'end'
            self.checkpoint()
        if self.wbox and self.rank == 0:
            target = getattr(self.model, 'GB' if self.kind == 'translation' else 'G')
            ber = float(self.model.loss_model.compute_ber(target))
            with open(os.path.join(self.config.log.path, 'metrics.json'), 'w') as f:
                from iprgan import _lib
                json.dump({'BER': ber, 'engine': {**self.engine, 'math': _lib.get_math()}}, f)
        return last


def seed_all(seed):
    torch.manual_seed(seed)
    np.random.seed(seed)
    random.seed(seed)


def main():
    ap = argparse.ArgumentParser(description='Training script (HIP engine)')
    ap.add_argument('-c', '--config', required=True, help='path to a reference config file')
    ap.add_argument('--max-steps', type=int, default=None, help='stop after this many iterations (smoke runs)')
    ap.add_argument('--log-path', default=None, help='override log.path')
    args = ap.parse_args()
    config = Config.parse(args.config)
    if args.log_path:
        config.log.path = args.log_path
    seed_all(config.seed)                                       # train.py:43-47
    exp = Experiment(config)
    # Construction used the SAME seed on every rank on purpose: the black-box trigger / target modules
    # (RandomBitMask, RandomNoisePatch, TransformVar) and a random sign string are drawn there and must be identical
    # across replicas.  From here on every rank draws its OWN data shard, latents and ImagePool decisions
    # ("each rank draws its own z (seed + rank)").
    seed_all(config.seed + exp.rank)
    ckpt = os.path.join(config.log.path, 'checkpoint.pt')
    if os.path.exists(ckpt):                                    # train.py:26-31 auto-resume
        exp.load_state_dict(torch.load(ckpt, map_location=exp.device[0]))
    if os.environ.get('IPRGAN_TRAIN_PROBE'):                    # test hook: what this rank is about to train on
        x = next(exp.data_loader)[0]
        p = torch.cat([t.detach().flatten()[:64].cpu() for t in exp.model.G.parameters()])
        torch.save({'rank': exp.rank, 'batch_sum': float(x.double().sum()), 'param_probe': p},
                   os.environ['IPRGAN_TRAIN_PROBE'] + f'.{exp.rank}')
    exp.start(max_steps=args.max_steps)
    from iprgan import parallel
    if dist.is_initialized() and not parallel.RcclTransport.abandoned:
        dist.destroy_process_group()
    parallel.finish(0)          # (os._exit after an abandoned RCCL bring-up: a helper thread is still inside the library)


if __name__ == '__main__':
    main()
