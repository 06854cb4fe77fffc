"""GPU: the thin train.py driver runs a reference-format YAML, checkpoints in the reference layout and resumes."""
import json
import os
import subprocess
import sys

import pytest
import torch

from conftest import PKG, ROOT

pytestmark = pytest.mark.gpu


def test_train_driver_checkpoint_and_resume(tmp_path):
    cfg = os.path.join(ROOT, 'tests', 'configs', 'dcgan-wbox-tiny.yaml')
    log = str(tmp_path / 'log')
    cmd = [sys.executable, os.path.join(PKG, 'train.py'), '-c', cfg, '--log-path', log]
    subprocess.run(cmd + ['--max-steps', '3'], check=True, timeout=600)
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 2 and list(sd) == ['G', 'D', 'optG', 'optD', 'sign', 'step']
    assert 'module.convs.0.1.weight' in sd['G'] and 'module.net.0.0.weight_orig' in sd['D']
    subprocess.run(cmd, check=True, timeout=600)               # auto-resume at step 3, runs to the end
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 'END'
    steps = [json.loads(l)['step'] for l in open(os.path.join(log, 'metrics.jsonl'))]
    assert steps == [1, 2, 3, 3, 4]                             # step 3 ran twice: before and after the resume
    assert json.load(open(os.path.join(log, 'metrics.json')))['BER'] == 0.0
    # the checkpoint loads into the CPU oracle (reference layout)
    from oracle import cases, gan
    o = gan.WhiteBoxWrapper(gan.DCGAN(gan.Cfg(cases.DCGAN_CFG)), gan.Cfg(cases.WBOX_CFG))
    o.load_state_dict({k: v for k, v in sd.items() if k != 'step'}, strict=True)


def test_train_driver_vae(tmp_path):
    """configs/VAE/*: same ImageGeneration experiment, models.VAE with one optimizer stored under 'opt'."""
    cfg = os.path.join(ROOT, 'tests', 'configs', 'vae-wbox-tiny.yaml')
    log = str(tmp_path / 'log')
    subprocess.run([sys.executable, os.path.join(PKG, 'train.py'), '-c', cfg, '--log-path', log], check=True, timeout=600)
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 'END' and list(sd) == ['G', 'D', 'opt', 'sign', 'step']
    assert 'module.3.weight' in sd['G'] and 'module.q_logvar.bias' in sd['D']
    rows = [json.loads(l) for l in open(os.path.join(log, 'metrics.jsonl'))]
    assert [r['step'] for r in rows] == [1, 2, 3] and all(r['G/KL'] > 0 and r['G/R'] > 0 for r in rows)
    assert json.load(open(os.path.join(log, 'metrics.json')))['BER'] == 0.0
    from oracle import cases, gan
    o = gan.WhiteBoxWrapper(gan.VAE(gan.Cfg(cases.VAE_CFG)), gan.Cfg(cases.WBOX_CFG))
    o.load_state_dict({k: v for k, v in sd.items() if k != 'step'}, strict=True)


def test_train_driver_complete_protection(tmp_path):
    """configs/DCGAN/complete: black-box (trigger set + SSIM) and white-box (sign loss) wrappers together."""
    cfg = os.path.join(ROOT, 'tests', 'configs', 'dcgan-complete-tiny.yaml')
    log = str(tmp_path / 'log')
    subprocess.run([sys.executable, os.path.join(PKG, 'train.py'), '-c', cfg, '--log-path', log], check=True, timeout=600)
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert list(sd) == ['G', 'D', 'optG', 'optD', 'fn_inp', 'fn_out', 'sign', 'step'] and sd['step'] == 'END'
    assert list(sd['fn_out']) == ['module.bg', 'module.fg']
    rows = [json.loads(l) for l in open(os.path.join(log, 'metrics.jsonl'))]
    assert len(rows) == 3 and all(0.0 < r['P/SSIM'] <= 1.0 and 'P/SignLoss' in r for r in rows)
    assert json.load(open(os.path.join(log, 'metrics.json')))['BER'] == 0.0


def test_train_driver_engine_block(tmp_path):
    """The optional `engine:` block of the YAML (iprgan/configs.py): bf16 MFMA tiles with bf16 activations selected from
    the config file; the run records the mode it trained in and still embeds the watermark."""
    base = open(os.path.join(ROOT, 'tests', 'configs', 'dcgan-wbox-tiny.yaml')).read()
    cfg = tmp_path / 'bf16.yaml'
    cfg.write_text(base.replace('iteration: 4', 'iteration: 6') + "engine:\n  math: 'bf16act'\n  bucket_mb: 4\n  graph: true\n")
    log = str(tmp_path / 'log')
    subprocess.run([sys.executable, os.path.join(PKG, 'train.py'), '-c', str(cfg), '--log-path', log], check=True, timeout=600)
    m = json.load(open(os.path.join(log, 'metrics.json')))
    assert m['engine'] == {'math': 'bf16act', 'bucket_mb': 4, 'graph': True} and m['BER'] == 0.0
    steps = [json.loads(l)['step'] for l in open(os.path.join(log, 'metrics.jsonl'))]
    assert steps == [1, 2, 3, 4, 5, 6]           # (three eager warm-up calls, the capture + its replay, two more replays)
    bad = tmp_path / 'bad.yaml'
    bad.write_text(base + "engine:\n  maths: 'bf16'\n")
    r = subprocess.run([sys.executable, os.path.join(PKG, 'train.py'), '-c', str(bad), '--log-path', log + '2'],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and 'unknown key' in r.stderr


def test_two_ranks_draw_different_data_and_hold_equal_parameters(tmp_path):
    """train.py seeded every rank identically.  Two ranks (sharing the test box's one GPU, gloo) now build
    the model with the common seed - identical parameters, sign buffers and trigger modules - and draw their data /
    latents / ImagePool decisions from seed + rank: the first batches differ, the replicas are equal before and after
    training (the reduced gradients are the same on both)."""
    cfg = os.path.join(ROOT, 'tests', 'configs', 'dcgan-wbox-tiny.yaml')
    log, probe = str(tmp_path / 'log'), str(tmp_path / 'probe')
    env = dict(os.environ, IPRGAN_SHARE_DEVICE='1', IPRGAN_DIST_BACKEND='gloo', IPRGAN_TRAIN_PROBE=probe)
    subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                    '127.0.0.1', '--master-port', '29517', os.path.join(PKG, 'train.py'), '-c', cfg, '--log-path', log],
                   check=True, timeout=900, env=env)
    a, b = torch.load(probe + '.0'), torch.load(probe + '.1')
    assert a['batch_sum'] != b['batch_sum'], 'both ranks drew the same first batch'
    assert torch.equal(a['param_probe'], b['param_probe']), 'replicas start from different parameters'
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 'END'              # iteration 4 // world 2 = 2 steps per rank
    assert json.load(open(os.path.join(log, 'metrics.json')))['BER'] == 0.0


def test_train_driver_srgan_pretrain_then_gan_phase(tmp_path):
    """configs/SRGAN/*: ImageSuperResolution.train() (experiments/image_super_resolution.py:84-113) through the driver -
    `pretrain_iter` content-loss-only generator steps (no discriminator update, black-box inhibited), then the GAN phase in
    the reference's order (update_g, then update_d on the tensors update_g left), both learning rates x 0.1 at
    pretrain_iter + iteration // 2; checkpoint in the reference layout, loadable by the CPU oracle."""
    cfg = os.path.join(ROOT, 'tests', 'configs', 'srgan-wbox-tiny.yaml')
    log = str(tmp_path / 'log')
    subprocess.run([sys.executable, os.path.join(PKG, 'train.py'), '-c', cfg, '--log-path', log], check=True, timeout=900)
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 'END' and list(sd) == ['G', 'D', 'optG', 'optD', 'sign', 'step']
    rows = [json.loads(l) for l in open(os.path.join(log, 'metrics.jsonl'))]
    assert [r['step'] for r in rows] == [1, 2, 3, 4, 5, 6]                 # pretrain_iter 2 + iteration 4
    # pretraining (models/srgan.py:76-79): pixel MSE only - no perceptual / adversarial term, the discriminator has not run
    for r in rows[:2]:
        assert r['G/MSE'] > 0 and r['G/Sum'] == r['G/MSE'] and 'G/Adv' not in r and 'D/Sum' not in r, r
    for r in rows[2:]:
        assert r['G/Con'] > 0 and r['G/Adv'] > 0 and r['D/Sum'] > 0 and r['D/Real'] > 0 and r['D/Fake'] > 0, r
    # lr *= 0.1 on both optimizers at step pretrain_iter + iteration // 2 = 4 (image_super_resolution.py:88-90)
    assert abs(sd['optG']['param_groups'][0]['lr'] - 1e-5) < 1e-12 and abs(sd['optD']['param_groups'][0]['lr'] - 1e-5) < 1e-12
    # Adam step counts: G stepped in all six iterations, D in the four of the GAN phase
    assert {int(v['step']) for v in sd['optG']['state'].values()} == {6}
    assert {int(v['step']) for v in sd['optD']['state'].values()} == {4}
    assert json.load(open(os.path.join(log, 'metrics.json')))['BER'] == 0.0
    from oracle import cases, gan
    o = gan.WhiteBoxWrapper(gan.SRGAN(gan.Cfg(cases.SRGAN_CFG)), gan.Cfg(cases.WBOX_CFG))
    o.load_state_dict({k: v for k, v in sd.items() if k != 'step'}, strict=True)


def test_train_driver_cyclegan_epochs_pools_and_lr_decay(tmp_path):
    """configs/CycleGAN/*: ImageTranslation.train() (experiments/image_translation.py:90-112) through the driver - iteration
    and log.freq are EPOCHS (3 synthetic samples at batch 1 = 3 iterations each), update_lr() at the first iteration of every
    epoch but the first, the LambdaLR rule of models/cyclegan.py:50-51 over model.epoch = 4 (1, 1, 1, 0.5), update_g before
    update_d, image pools (size 2) past their fill; checkpoint in the reference layout (schedulers and pools included),
    loadable by the CPU oracle; interrupted after 5 iterations and resumed."""
    cfg = os.path.join(ROOT, 'tests', 'configs', 'cyclegan-wbox-tiny.yaml')
    log = str(tmp_path / 'log')
    cmd = [sys.executable, os.path.join(PKG, 'train.py'), '-c', cfg, '--log-path', log]
    subprocess.run(cmd + ['--max-steps', '5'], check=True, timeout=900)
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 3                                   # log.freq = 1 epoch = 3 iterations: the checkpoint of epoch 1
    subprocess.run(cmd, check=True, timeout=900)             # resumes at iteration 4
    sd = torch.load(os.path.join(log, 'checkpoint.pt'), map_location='cpu')
    assert sd['step'] == 'END'
    assert list(sd)[:4] == ['GA', 'GB', 'DA', 'DB'] and {'optG', 'optD', 'sign'} <= set(sd)
    rows = [json.loads(l) for l in open(os.path.join(log, 'metrics.jsonl'))]
    assert [r['step'] for r in rows] == [1, 2, 3, 4, 5] + list(range(4, 13))
    lr = {r['step']: r['LR'] for r in rows[5:]}
    assert all(abs(lr[s] - 2e-4) < 1e-12 for s in range(4, 10)), lr           # epochs 2, 3: factor 1
    assert all(abs(lr[s] - 1e-4) < 1e-12 for s in range(10, 13)), lr          # epoch 4: 1 - (3 - 2) / 2
    for r in rows:
        assert all(r[k] > 0 for k in ('G/A', 'G/B', 'G/CycA', 'G/CycB', 'G/IdtA', 'G/IdtB', 'D/SumA', 'D/SumB')), r
    assert json.load(open(os.path.join(log, 'metrics.json')))['BER'] == 0.0
    from oracle import cases, gan
    o = gan.WhiteBoxWrapper(gan.CycleGAN(gan.Cfg(dict(cases.CYCLEGAN_CFG, G='Resnet9Blocks', pool_size=2, epoch=4))),
                            gan.Cfg(dict(cases.WBOX_CFG, target='GB')))
    o.load_state_dict({k: v for k, v in sd.items() if k != 'step'}, strict=True)
