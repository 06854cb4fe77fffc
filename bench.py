"""bench.py — imgs/sec of one G+D training step on the HIP engine.

Contract (driver): ``python bench.py --gpus N --steps K --warmup W``; for N>1 it is launched by
``python -m torch.distributed.run --nproc-per-node N ...`` (one process per GPU, RCCL).  Rank 0
prints ONE JSON line.  Started WITHOUT a launcher (no WORLD_SIZE in the environment) and ``--gpus N`` > 1, this process
starts that launcher itself as a child - before it touches the GPU - relays rank 0's line and exits with the child's code
(per-GPU batch fixed, one replica per process: experiments/base.py:36-39 without the DataParallel scatter).
The default workload is the headline metric of BASELINE.json:

    DCGAN-64 + sign-loss white-box watermark, batch 128 per GPU, fp32          (--workload dcgan64)

A "step" is the body of the reference's hot loop, ``ImageGeneration.train()``
(experiments/image_generation.py:86-101): ``model.update_d({real_sample, latent})`` then
``model.update_g({fake_sample: model.fake_sample})`` with ``models.WhiteBoxWrapper`` on top
(configs/DCGAN: gamma_0 0.1, string 'EXAMPLE A'), Adam(2e-4, (0.5, 0.999)) on G and D.
Inputs are synthetic (x = tanh(randn), z = randn) and already resident in HBM
when the timed region starts; weights are random-init.  Nothing is skipped: both optimizer steps,
spectral-norm power iterations, BatchNorm statistics and the sign loss run inside the timed region.

``--workload`` selects the other BASELINE.json configs with the same JSON schema (secondary lines, not the headline):
    dcgan128  DCGAN 128x128 batch 256 (config 5; add --math bf16 for its bf16 MFMA tiles)
    srgan     SRGAN 24->96 GAN-phase step, batch 64, VGG19 features (image_super_resolution.py:84-113)
    cyclegan  CycleGAN Resnet9Blocks + PatchGAN, 256x256, batch 8 per GPU (image_translation.py:90-112)

Extra objects on the JSON line:
  roofline      the conv kernel with the largest share of device time: algorithmic FLOPs
                (2*B*OH*OW*Cout*Cin*KH*KW per launch) / HIP-event launch durations measured live in this run, against
                the MFMA peak of the math mode (157.3 TFLOP/s fp32, 2500 bf16; MI355X_MICROARCH.md).  ``traffic`` and
                ``mfma_util_pct_pmc`` are NOT measured in this run: they are read from the committed rocprofv3 --pmc
                passes under profiles/ and carry their file name in ``*_source``.
  step_roofline SURVEY.md 8(d)'s bounding roofline of the whole step: sum over the step's entry-point calls of
                max(2*MACs / MFMA peak of the math mode, operand + result bytes / 6.3 TB/s), from a host-side accounting of one
                eager step (iprgan/_lib.py: acct_*); frac = bound_ms / ms_per_step
  north_star_conv  BASELINE.json's target layer (3x3 256->256 reflect-padded, 64x64 maps, batch 64) forward / backward-data /
                backward-weight in the run's math mode and on the exact fp32 MFMA, timed after the timed region
  eager_ms_per_step / host_enqueue_ms_per_eager_step  device and host cost of a step enqueued kernel by kernel (what N > 1 runs
                under the default `--graph auto`), from a window behind the timed region
  cpu_baseline  the CPU oracle (oracle/gan.py, plain PyTorch fp32: the reference's own arithmetic) timed on this box's
                host cores on a bounded sample of the same workload (rank 0, N=1), BEFORE the GPU section
"""
import argparse
import gc
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'ipr-gan_amd')):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP32_MFMA = 157.3e12
PEAK_BF16_MFMA = 2500e12          # dense bf16 MFMA (MI355X_MICROARCH.md); only used with --math bf16
# --math fp32x3: an fp32 product block is SIX bf16 MFMAs (operands split into three bf16 terms, conv_igemm.hip SPLIT), so
# the matrix pipe bounds the fp32-equivalent rate at a sixth of the dense bf16 peak
PEAK_X3_MFMA = PEAK_BF16_MFMA / 6
PROF_STEPS = int(os.environ.get('IPRGAN_BENCH_PROF_STEPS', '0'))    # eager, instrumented steps AFTER the timed region (0 = K / 4, 2 ... 25)
WBOX_CFG = {'gamma_0': 0.1, 'string': 'EXAMPLE A'}
ADAM_GAN = {'lr': 2.0e-4, 'betas': [0.5, 0.999]}


def _dcgan_cfg(size):
    return {'G': f'ConvGenerator{size}', 'D': f'SNDiscriminator{size}', 'opt': 'Adam', 'opt_param': ADAM_GAN,
            'type': 'DCGAN'}


DCGAN_CFG = _dcgan_cfg(64)        # (kept under this name: tests and scripts import it)
SRGAN_CFG = {'G': 'SRResNet', 'D': 'Discriminator96', 'V': 'VGG19Feature', 'opt': 'Adam', 'opt_param': {'lr': 1.0e-4},
             'type': 'SRGAN'}
CYCLEGAN_CFG = {'G': 'Resnet9Blocks', 'D': 'ConvDiscriminator', 'opt': 'Adam', 'opt_param': ADAM_GAN, 'type': 'CycleGAN',
                'pool_size': 50, 'lambda_A': 10.0, 'lambda_B': 10.0, 'lambda_idt': 0.5, 'epoch': 200}

# per-sample algorithmic GFLOP of one step (SURVEY.md section 8d / BASELINE.md section 4: 2*MAC of conv / conv-transpose /
# linear layers, forward + the backward passes whose gradient is consumed)
WORKLOADS = {
    'dcgan64': dict(batch=128, gflop=3 * 0.8279 + 8 * 0.8699, unit='img/s',
                    metric='imgs/sec G+D step (DCGAN-64 bs128)',
                    text='DCGAN-64 (ConvGenerator64 + SNDiscriminator64) + sign-loss white-box watermark, G+D step, '
                         'batch 128 per GPU'),
    'dcgan128': dict(batch=256, gflop=3 * 3.3114 + 8 * 3.4794, unit='img/s',
                     metric='imgs/sec G+D step (DCGAN-128 bs256)',
                     text='DCGAN-128 (ConvGenerator(mg=16) + SNDiscriminator(md=16)) + sign-loss white-box watermark, '
                          'G+D step, batch 256 per GPU'),
    'srgan': dict(batch=64, gflop=3 * 2.5553 + 8 * 1.7683 + 3 * 7.1664, unit='img/s',
                  metric='imgs/sec G+D step (SRGAN 24->96 bs64, GAN phase)',
                  text='SRGAN GAN phase (SRResNet + Discriminator96 + VGG19[:36] features, random-init) + sign-loss '
                       'watermark, G+D step, 24x24 -> 96x96, batch 64 per GPU'),
    'cyclegan': dict(batch=8, gflop=18 * 99.103 + 16 * 6.2936, unit='pairs/s',
                     metric='image pairs/sec G+D step (CycleGAN Resnet9 256x256 bs8)',
                     text='CycleGAN (2 x Resnet9Blocks + 2 x ConvDiscriminator) + sign-loss watermark on GB, G+D step, '
                          '256x256, batch 8 per GPU'),
}


def make_workload(name, device, impl):
    """Build the model and the step closure of a workload on ``device`` with implementation ``impl`` = (Config class,
    models namespace): the HIP engine, or the CPU oracle for the baseline.  Returns (model, step, metrics_fn)."""
    make_cfg, models = impl
    w = WORKLOADS[name]
    B = w['batch']
    dev = device[0]
    if name in ('dcgan64', 'dcgan128'):
        S = 64 if name == 'dcgan64' else 128
        model = models.WhiteBoxWrapper(models.DCGAN(make_cfg(_dcgan_cfg(S)), device=device), make_cfg(dict(WBOX_CFG, target='G')))
        pool = 8 if S == 64 else 2                      # synthetic batches resident on the device, cycled
        xs = [torch.tanh(torch.randn(B, 3, S, S, device=dev)) for _ in range(pool)]
        zs = [torch.randn(B, 128, device=dev) for _ in range(pool)]

        def step(i):
            model.update_d({'real_sample': xs[i % pool], 'latent': zs[i % pool]})
            model.update_g({'fake_sample': model.fake_sample})

        def graph_body(s):          # the same step on static inputs (iprgan/graphs.py)
            model.update_d({'real_sample': s['x'], 'latent': s['z']})
            model.update_g({'fake_sample': model.fake_sample})
        step.graph_spec = (graph_body, lambda i: {'x': xs[i % pool], 'z': zs[i % pool]})
    elif name == 'srgan':
        model = models.WhiteBoxWrapper(models.SRGAN(make_cfg(SRGAN_CFG), device=device), make_cfg(dict(WBOX_CFG, target='G')))
        pool = 4
        lrs = [torch.rand(B, 3, 24, 24, device=dev) for _ in range(pool)]
        hrs = [torch.rand(B, 3, 96, 96, device=dev) for _ in range(pool)]

        def step(i):
            model.update_g({'low_res': lrs[i % pool], 'high_res': hrs[i % pool], 'pretrain': False})
            model.update_d({'high_res': model.high_res, 'super_res': model.super_res})

        def graph_body(s):
            model.update_g({'low_res': s['lr'], 'high_res': s['hr'], 'pretrain': False})
            model.update_d({'high_res': model.high_res, 'super_res': model.super_res})
        step.graph_spec = (graph_body, lambda i: {'lr': lrs[i % pool], 'hr': hrs[i % pool]})
    else:
        model = models.WhiteBoxWrapper(models.CycleGAN(make_cfg(CYCLEGAN_CFG), device=device), make_cfg(dict(WBOX_CFG, target='GB')))
        pool = 2
        As = [torch.tanh(torch.randn(B, 3, 256, 256, device=dev)) for _ in range(pool)]
        Bs = [torch.tanh(torch.randn(B, 3, 256, 256, device=dev)) for _ in range(pool)]

        def step(i):
            model.update_g({'real_A': As[i % pool], 'real_B': Bs[i % pool]})
            model.update_d({'real_A': model.real_A, 'real_B': model.real_B,
                            'fake_A': model.fake_A.detach(), 'fake_B': model.fake_B.detach()})

        def graph_body(s):          # capturable once both image pools are full (7 steps at batch 8, pool 50): models.CycleGAN.graph_*
            model.update_g({'real_A': s['a'], 'real_B': s['b']})
            model.update_d({'real_A': model.real_A, 'real_B': model.real_B,
                            'fake_A': model.fake_A.detach(), 'fake_B': model.fake_B.detach()})
        step.graph_spec = (graph_body, lambda i: {'a': As[i % pool], 'b': Bs[i % pool]})
    return model, step


def build_model(device):
    """The headline model on ``device`` (used by scripts/ and tests)."""
    from iprgan import Config, models
    return models.WhiteBoxWrapper(models.DCGAN(Config(DCGAN_CFG), device=[device]), Config(dict(WBOX_CFG, target='G')))


def step(model, x, z):
    model.update_d({'real_sample': x, 'latent': z})
    model.update_g({'fake_sample': model.fake_sample})


def log(msg):
    print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def host_cpu():
    """CPU model / logical CPUs / physical cores of this box (SURVEY.md section 8d asks for them next to the baseline)."""
    model, phys = 'unknown', set()
    try:
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name') and model == 'unknown':
                model = line.split(':', 1)[1].strip()
            elif line.startswith('physical id'):
                pid = line.split(':', 1)[1].strip()
            elif line.startswith('core id'):
                cid = line.split(':', 1)[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    return {'model': model, 'logical_cpus': os.cpu_count(), 'physical_cores': len(phys) or None}


def cpu_baseline(name, warm=3, max_timed=10, budget_s=30.0):
    """The oracle's step on the host CPU: ``warm`` warm-up + up to ``max_timed`` timed steps of the same workload,
    bounded by ``budget_s`` seconds of timed work (at least one timed step; the warm-up is cut to one step when a
    single step already takes more than a third of the budget).  SURVEY.md section 8d protocol: 3 warm-up + >= 10 timed
    where the budget allows; the sample actually taken is stated in the result.  Thread count: torch's default (every
    logical CPU) oversubscribes a 128-thread host on these small convolutions, so when a step is short enough a two-step
    probe of 32 / 64 / all threads picks the fastest setting first; the probe is reported in ``sample``."""
    from oracle import gan
    torch.manual_seed(1234)
    all_threads = threads = torch.get_num_threads()            # torch's default = cores this process may use
    w = WORKLOADS[name]
    if name == 'dcgan64' and threads > 32:
        # the headline's small convolutions: the first (cold) step with every logical CPU took > 12 s on one box, which
        # skipped the probe below and left the baseline oversubscribed (11.7 instead of 50-64 img/s): start moderate
        threads = 32
        torch.set_num_threads(threads)
    model, step_fn = make_workload(name, gan.CPU, (gan.Cfg, gan))
    t0 = time.perf_counter()
    step_fn(0)
    first = time.perf_counter() - t0
    n_warm = 1
    probe = ''
    if first < 12.0:
        cands = sorted({t for t in (16, 32, 64, all_threads) if 1 <= t <= all_threads})
        best_t, best_dt, seen = threads, None, []
        for t in cands:
            torch.set_num_threads(t)
            step_fn(n_warm)
            t1 = time.perf_counter()
            step_fn(n_warm + 1)
            step_fn(n_warm + 2)
            d = (time.perf_counter() - t1) / 2
            n_warm += 3
            seen.append(f'{t}: {d * 1e3:.0f} ms')
            if best_dt is None or d < best_dt:
                best_t, best_dt = t, d
        threads = best_t
        torch.set_num_threads(threads)
        first = best_dt
        probe = f'; thread probe (2 steps each) {", ".join(seen)} -> {threads}'
    while n_warm < warm and first * 3 < budget_s:
        step_fn(n_warm)
        n_warm += 1
    log(f'cpu baseline: {n_warm} warm-up step(s), first {first:.1f}s, {threads} threads')
    n_timed, t0 = 0, time.perf_counter()
    while n_timed < max_timed and (n_timed == 0 or time.perf_counter() - t0 < budget_s):
        step_fn(n_warm + n_timed)
        n_timed += 1
    dt = (time.perf_counter() - t0) / n_timed
    cpu = host_cpu()
    return {'value': round(w['batch'] / dt, 3), 'unit': w['unit'], 'cores': threads, 'kind': 'port',
            'cpu_model': cpu['model'], 'logical_cpus': cpu['logical_cpus'], 'physical_cores': cpu['physical_cores'],
            'sample': f'{n_timed} timed steps (+{n_warm} warm-up; budget {budget_s:.0f}s of timed work) of {w["text"]}, '
                      f'fp32, torch {torch.__version__} CPU, {threads} threads; {dt * 1e3:.0f} ms/step{probe}'}


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: run ``python -m torch.distributed.run`` with N ranks as a CHILD
    process (never an exec, and before this process initialises the GPU), let it inherit stdout - rank 0's JSON line is
    the only thing the ranks print there - and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f'no launcher in the environment: starting {n} ranks: {" ".join(cmd[1:8])} ...')
    return subprocess.call(cmd, env=env)


HBM_ACHIEVABLE = 6.3e12          # B/s: what a streaming kernel reaches of the 8 TB/s spec (MI355X_MICROARCH.md)


def step_roofline(recs, peak_mfma, ms_measured):
    """SURVEY.md section 8(d) "bounding roofline" of one step: sum over the step's entry-point calls of
    max(flops / peak_mfma, algorithmic bytes / achievable HBM rate).  ``recs`` = _lib.acct_end() of ONE eagerly enqueued step:
    flops = 2 * MACs of the convolution family, bytes = every operand and result of a call counted once (workspaces excluded)."""
    mfma = sum(f for _, f, _ in recs) / peak_mfma
    hbm = sum(b for _, _, b in recs) / HBM_ACHIEVABLE
    bound = sum(max(f / peak_mfma, b / HBM_ACHIEVABLE) for _, f, b in recs)
    mfma_bound_ops = sum(1 for _, f, b in recs if f / peak_mfma >= b / HBM_ACHIEVABLE and f > 0)
    by = {}
    for name, f, b in recs:
        e = by.setdefault(name.replace('iprgan_', ''), [0, 0.0, 0.0, 0.0])
        e[0] += 1; e[1] += f; e[2] += b; e[3] += max(f / peak_mfma, b / HBM_ACHIEVABLE)
    top = sorted(by.items(), key=lambda kv: -kv[1][3])[:8]
    return {'bound_ms': round(bound * 1e3, 4), 'mfma_ms': round(mfma * 1e3, 4), 'hbm_ms': round(hbm * 1e3, 4),
            'frac': round(bound * 1e3 / ms_measured, 4) if ms_measured else None,
            'peak_mfma_tflops': round(peak_mfma / 1e12, 1), 'hbm_tb_s': HBM_ACHIEVABLE / 1e12,
            'calls': len(recs), 'calls_mfma_bound': mfma_bound_ops,
            'algorithmic_gflop': round(sum(f for _, f, _ in recs) / 1e9, 2),
            'algorithmic_mb': round(sum(b for _, _, b in recs) / 1e6, 1),
            'by_entry_point': {k: {'calls': v[0], 'gflop': round(v[1] / 1e9, 2), 'mb': round(v[2] / 1e6, 1),
                                   'bound_ms': round(v[3] * 1e3, 4)} for k, v in top},
            'definition': 'sum over the calls of one step of max(2*MACs / peak_mfma, operand+result bytes / hbm_tb_s); '
                          'frac = bound_ms / ms_per_step'}


def layer_bytes(tag):
    """Algorithmic bytes of one launch of a conv-family layer from its profiler tag ("fwd B256 64x64 64->64 k4x4 s2 p1 x2 y2":
    pass, geometry, storage kinds of the input / output side): each operand and the result once - activations at 4 / 2 / 6 bytes
    per element of their padded channel count (fp32 / bf16 / three planes), weights fp32 for backward-weight and in the operand's
    kind otherwise."""
    import re
    m = re.match(r'(\w+)\s+B(\d+) (\d+)x(\d+) (\d+)->(\d+) k(\d+)x(\d+) s(\d+) p(\d+)( T)?( reflect)? x(\d) y(\d)', tag)
    if not m:
        return 0.0
    ps, B, H, W, cin, cout, kh, kw, st, pad = m.group(1), *[int(m.group(i)) for i in range(2, 11)]
    tr, xk, yk = bool(m.group(11)), int(m.group(13)), int(m.group(14))
    if tr:
        OH, OW = (H - 1) * st - 2 * pad + kh, (W - 1) * st - 2 * pad + kw
    else:
        OH, OW = (H + 2 * pad - kh) // st + 1, (W + 2 * pad - kw) // st + 1
    esz = {0: 4, 1: 2, 2: 6}
    c4 = lambda c: (c + 3) & ~3          # noqa: E731
    xb, yb = B * H * W * c4(cin) * esz[xk], B * OH * OW * c4(cout) * esz[yk]
    wn = cin * cout * kh * kw
    if ps == 'wgrad':
        return float(xb + yb + 4 * wn)
    wb = wn * esz[xk if ps == 'fwd' else yk]
    return float(xb + yb + wb)


def north_star_conv(device, modes, n=8):
    """BASELINE.json north_star microbench: the 3x3 256->256 stride-1 reflection-padded convolution of Resnet9Blocks on a
    64x3x256x256 batch (networks/resnet_generator.py:44-49: 64x64 maps, batch 64), forward / backward-data /
    backward-weight through the C ABI, per math mode: mean of ``n`` launches behind two untimed ones (the first autotunes),
    device time from events on the launch stream.  Runs AFTER the timed region; never part of `value`."""
    from iprgan import _lib, ops
    out = {}
    keep = _lib.get_math()
    B, H, Cc = 64, 64, 256
    for mode in modes:
        _lib.set_math(mode)
        peak = {'fp32': PEAK_FP32_MFMA, 'fp32x3': PEAK_X3_MFMA}.get(mode, PEAK_BF16_MFMA)
        spec = ops.ConvSpec(Cc, Cc, 3, 1, 1, pad_mode=1)
        d = spec.desc(B, H, H)
        g = torch.Generator(device='cpu').manual_seed(7)
        x = ops.to_kind(torch.randn(B, H, H, Cc, generator=g).to(device), d.x_bf16)
        dy = ops.to_kind(torch.randn(B, H, H, Cc, generator=g).to(device), d.y_bf16)
        w = (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.05).to(device)
        wf, wb = ops.conv_prep(spec, d, w, None, True, True)
        flops = ops.conv_flops(spec, d)
        res = {}
        for name, fn in (('fwd', lambda: ops.conv_fwd(spec, d, x, wf, None)),
                         ('dgrad', lambda: ops.conv_bwd_data(spec, d, dy, wb)),
                         ('wgrad', lambda: ops.conv_bwd_weight(spec, d, x, dy, tuple(w.shape), False))):
            fn(); fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / n
            res[name] = {'ms': round(ms, 4), 'tflops': round(flops / ms / 1e9, 1), 'frac': round(flops / (ms * 1e-3) / peak, 4)}
        res['peak_tflops'] = round(peak / 1e12, 1)
        out[mode] = res
        del x, dy, w, wf, wb
    _lib.set_math(keep)
    out['shape'] = f'Conv2d 256->256 k3 s1 ReflectionPad2d(1), {H}x{H} maps, batch {B} (the residual-block conv of Resnet9Blocks on 64x3x256x256); {n} launches each'
    return out


def _latest_profile(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
    return files[-1] if files else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None)
    ap.add_argument('--warmup', type=int, default=None)
    ap.add_argument('--workload', choices=list(WORKLOADS), default='dcgan64')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--alt-math', choices=['fp32', 'fp32x3', 'auto', 'none'], default='auto',
                    help="after the timed region of a single-GPU run in one of the two fp32 modes, time the same steps once "
                         "more in the OTHER one and report them under 'alt_math' (never in 'value'); 'auto' = the other mode")
    ap.add_argument('--math', choices=['fp32', 'bf16', 'bf16act', 'fp32x3'], default='fp32x3',
                    help="conv math mode.  Both 'fp32x3' (default: fp32 tensors stored as three exact bf16 planes, products "
                         "from six bf16 MFMAs per block with fp32 accumulation - every GPU parity test runs in this mode at "
                         "the fp32 tolerances, per-layer error against float64 at or below the fp32 MFMA's) and 'fp32' (the "
                         "exact fp32 MFMA, printed next to it under 'alt_math') are fp32 arithmetic.  'bf16' = bf16 MFMA "
                         "tiles with fp32 accumulation / master weights, reported with dtype bf16; 'bf16act' additionally "
                         "keeps activations with a multiple of 64 channels as bf16 in HBM")
    ap.add_argument('--graph', choices=['auto', 'on', 'off'], default='auto',
                    help="capture the whole step in one HIP graph (iprgan/graphs.py): 'auto' = where the step is "
                         "capturable (all workloads; N > 1: opt-in with 'on', through the C ABI's communicator); the per-kernel timer's steps run eagerly AFTER the timed region")
    ap.add_argument('--north-star', choices=['on', 'off'], default='on',
                    help="after the timed region of a single-GPU fp32 / fp32x3 run, time the north-star convolution (3x3 256->256 "
                         "reflect-padded, 64x64 maps, batch 64: forward / backward-data / backward-weight) in the run's math mode "
                         "and on the exact fp32 MFMA and report it under 'north_star_conv'")
    ap.add_argument('--clock-warm-ms', type=float, default=float(os.environ.get('IPRGAN_BENCH_CLOCK_WARM_MS', '0')),
                    help="keep the GPU busy for this long with convolutions on SCRATCH tensors (no model state touched) right before "
                         "the timed region: the chip needs ~0.2 s under load to settle its clocks (a 20-step window behind 5 warm-up steps "
                         "measures 2 %% slower than the same steps behind 20); default 0 = off, the W warm-up steps are all that runs")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    heavy = args.workload in ('cyclegan', 'dcgan128')
    if args.steps is None:                  # >= 20 warm-up, >= 100 timed where a step is milliseconds
        args.steps = {'cyclegan': 20, 'dcgan128': 10}.get(args.workload, 100)
    if args.warmup is None:             # (CycleGAN: its image pools fill during the first 7 steps; the step is captured after that)
        args.warmup = {'cyclegan': 10, 'dcgan128': 4}.get(args.workload, 20)
    if args.alt_math == 'auto':
        args.alt_math = {'fp32': 'fp32x3', 'fp32x3': 'fp32'}.get(args.math, 'none')

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus))     # nothing in this process has touched the GPU yet
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the HIP engine has no CPU path)')
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}')

    # The CPU baseline runs FIRST (rank 0, single-GPU runs only) so that the GPU section is the last thing this process
    # does: the driver's GPU-activity sampler then sees the timed region instead of a CPU-bound tail.
    baseline = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        baseline = cpu_baseline(args.workload, budget_s=20.0 if heavy else (60.0 if args.workload == 'dcgan64' else 30.0))

    ndev = torch.cuda.device_count()
    if local >= ndev and os.environ.get('IPRGAN_SHARE_DEVICE') == '1':
        local = local % ndev            # test-only: several ranks on one GPU (needs IPRGAN_DIST_BACKEND=gloo)
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        backend = os.environ.get('IPRGAN_DIST_BACKEND', 'nccl')     # nccl = RCCL over xGMI
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    from iprgan import Config, _lib, models
    _lib.set_math(args.math)
    torch.manual_seed(1234 + rank)              # every rank its own data shard / latents
    model, step_fn = make_workload(args.workload, [device], (Config, models))
    graphed = None
    # (auto: only when the capture - two eager calls, then the capturing one - fits inside the warm-up)
    # (N > 1: capture is opt-in, `--graph on` - the exchange is then captured with the step when it goes through the C ABI's
    # communicator; GraphedStep stays eager on every rank - `failed` says why - otherwise)
    if args.graph != 'off' and hasattr(step_fn, 'graph_spec') and (args.graph == 'on' or args.warmup >= 4):
        from iprgan import graphs
        body, inputs_of = step_fn.graph_spec
        graphed = graphs.GraphedStep(model, body, inputs_of(0), warmup=max(2, args.warmup - 2), allow_ddp=(args.graph == 'on'))
        eager_step = step_fn

        def step_fn(i, eager=False):              # noqa: F811  (the instrumented steps carry HIP events: they cannot be replayed)
            graphed(inputs_of(i), eager=eager)
    elif args.graph == 'on':
        raise SystemExit(f'bench.py: --graph on is not available for {args.workload}')

    log(f'{args.workload}: model built on {device}; warm-up {args.warmup} steps')
    for i in range(args.warmup):
        if i == max(0, args.warmup - 2):
            # Everything alive by now (torch's import graph, the model, the kernel tables, the captured step) lives for the whole
            # run: move it out of the cyclic collector's sight, so that a generation-2 pass (80-120 ms over ~1 M objects,
            # measured: one lands in any 50-step window) does not stall the enqueueing thread in the middle of the timed region.
            # train.py does the same.  The collection itself idles the GPU for ~80 ms and the clocks fall back to idle: it runs
            # BEFORE the last warm-up steps, not between them and the timed region (measured with `--steps 20 --warmup 5`:
            # 8.46-8.51 ms per step with the pause right in front of the timed region, against 8.24 in a 100-step run)
            torch.cuda.synchronize()
            gc.collect()
            gc.freeze()
        # the last warm-up step also warms the instrumentation (HIP event pool of the per-kernel timer)
        _lib.prof_enable(i == args.warmup - 1)
        if graphed is not None:
            step_fn(i, eager=(i == args.warmup - 1))
        else:
            step_fn(i)
        if world > 1 and i == 0:
            # every layer geometry of the step has been autotuned by now: all ranks adopt rank 0's choices, so that the
            # replicas run the same tiles (summation orders) whatever their own timings said
            from iprgan import parallel
            parallel.sync_autotune()
    _lib.prof_enable(False)
    _lib.prof_results()
    torch.cuda.synchronize()
    gc.freeze()                                  # (what the last warm-up steps left behind; no collection: no pause)
    if args.clock_warm_ms > 0:                   # opt-in, reported in the JSON line (`clock_warm_ms`)
        from iprgan import ops
        spec = ops.ConvSpec(256, 256, 3, 1, 1)
        dsc = spec.desc(16, 64, 64)
        xs = ops.to_kind(torch.randn(16, 64, 64, 256, device=device), dsc.x_bf16)
        wf, _ = ops.conv_prep(spec, dsc, torch.randn(256, 256, 3, 3, device=device) * 0.05, None, True, False)
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < args.clock_warm_ms:
            for _ in range(8):
                ops.conv_fwd(spec, dsc, xs, wf, None)
            torch.cuda.synchronize()
        del xs, wf
    log('warm-up done; timing')

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The timed region holds NOTHING but the K steps (replays of the captured step where the step is capturable): the
    # per-kernel HIP events of `roofline` / `conv_kernels` are taken in a SECOND window of eager steps behind the closing
    # fence (`value` no longer pays for its own instrumentation).
    # One device timestamp per step boundary (an event on the launch stream: torch's current stream is the one the library
    # launches on): median / p10 / p90 of the per-step device times next to the mean of the window.
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    _lib.prof_enable(False)
    fence()
    replays_before = graphed.replays if graphed is not None else 0
    t0 = time.perf_counter()
    stamps = []
    host_replayed, n_replayed = 0.0, 0                 # host time of the steps that went out as one graph launch
    marks[0].record()
    for i in range(args.steps):
        th = time.perf_counter()
        if graphed is not None:
            r0 = graphed.replays
            step_fn(i)
            if graphed.replays > r0:
                host_replayed += time.perf_counter() - th
                n_replayed += 1
        else:
            step_fn(i)
        marks[i + 1].record()
        stamps.append(time.perf_counter())
    host_elapsed = time.perf_counter() - t0          # the host has ENQUEUED all steps (no sync inside the loop)
    sclk = None
    if rank == 0:         # the shader clock WHILE the queued steps run (amdsmi through torch; the host has nothing else to do here)
        try:
            sclk = int(torch.cuda.clock_rate())
        except Exception:                             # noqa: BLE001
            sclk = None
    replays_timed = (graphed.replays - replays_before) if graphed is not None else 0
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    pct = lambda q: step_ms[min(len(step_ms) - 1, int(q * len(step_ms)))]          # noqa: E731
    log(f'timed {args.steps} steps in {elapsed:.3f}s')
    if os.environ.get('IPRGAN_BENCH_STAMPS'):
        log('per-step host ms: ' + ' '.join(f'{(b - a) * 1e3:.1f}' for a, b in zip([t0] + stamps[:-1], stamps)) +
            f' | final sync {(t0 + elapsed - stamps[-1]) * 1e3:.1f}')
    # eager window (behind the closing fence, never in `value`): the same steps enqueued kernel by kernel WITHOUT instrumentation -
    # what a step costs the device and the host when it is not a graph replay.  A data-parallel run under `--graph auto` is eager
    # in its timed region already (graphs.py: capture at N > 1 is opt-in), so these two fields are what makes the N = 1 line
    # comparable with the N > 1 lines.
    eager = None
    n_eager = max(2, min(10, args.steps))
    if graphed is not None or world == 1:
        em = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        torch.cuda.synchronize()
        te = time.perf_counter()
        em[0].record()
        for i in range(n_eager):
            if graphed is not None:
                step_fn(args.steps + i, eager=True)
            else:
                step_fn(args.steps + i)
        em[1].record()
        host_e = time.perf_counter() - te
        torch.cuda.synchronize()
        eager = {'ms': em[0].elapsed_time(em[1]) / n_eager, 'host_ms': host_e / n_eager * 1e3, 'steps': n_eager}
    # second window: the same steps enqueued kernel by kernel with a HIP-event pair on every conv-family dispatch
    prof_steps = PROF_STEPS if PROF_STEPS > 0 else max(2, min(25, args.steps // 4))
    _lib.prof_results()
    _lib.prof_enable(True)
    for i in range(prof_steps):
        if graphed is not None:
            step_fn(args.steps + i, eager=True)
        else:
            step_fn(args.steps + i)
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    kernels = [k for k in _lib.prof_results() if k['launches']]
    if os.environ.get('IPRGAN_BENCH_LAYERS') and rank == 0:     # per-layer table of the sampled steps, to stderr
        rows = sorted(_lib.prof_layers(), key=lambda r: -r['ms'])
        tot = sum(r['ms'] for r in rows)
        peak_l = {'fp32': PEAK_FP32_MFMA, 'fp32x3': PEAK_X3_MFMA}.get(args.math, PEAK_BF16_MFMA)
        bound_tot = 0.0
        log(f'conv-family layers, {prof_steps} sampled step(s), {tot / prof_steps:.3f} ms/step; bound = max(FLOPs / {peak_l / 1e12:.1f} TFLOP/s, '
            f'operand + result bytes / {HBM_ACHIEVABLE / 1e12:.1f} TB/s) per launch:')
        for r in rows:
            us = r['ms'] / r['launches'] * 1e3
            byt = layer_bytes(r['name'])
            bound = max(r['flops'] / r['launches'] / peak_l, byt / HBM_ACHIEVABLE) * 1e6 if byt else 0.0
            bound_tot += bound * r['launches'] / prof_steps
            log(f"  {r['name']:58s} n/step={r['launches'] / prof_steps:5.1f} us={us:8.1f} "
                f"ms/step={r['ms'] / prof_steps:7.3f} TF={r['flops'] / r['ms'] / 1e9 if r['ms'] else 0:6.1f} "
                f"bound_us={bound:7.1f} ({'hbm ' if byt / HBM_ACHIEVABLE > r['flops'] / r['launches'] / peak_l else 'mfma'}) x{us / bound if bound else 0:5.2f}")
        log(f'  sum of the layers\' bounds {bound_tot / 1e3:.3f} ms/step = {bound_tot / 1e3 / (tot / prof_steps):.3f} of their measured time')
    metrics = model.get_metrics()
    assert all(v == v for v in metrics.values()), f'non-finite metrics {metrics}'
    # accounting step: ONE more eager step with the host-side work accounting of _lib on (flops and operand / result bytes of
    # every entry-point call; nothing extra is launched) -> `step_roofline`
    _lib.acct_begin()
    try:
        if graphed is not None:
            step_fn(args.steps + prof_steps + n_eager, eager=True)
        else:
            step_fn(args.steps + prof_steps + n_eager)
    finally:
        acct = _lib.acct_end()
    torch.cuda.synchronize()

    # Second, separately reported measurement of the same workload in math mode 'fp32x3' (fp32 tensors, fp32-grade
    # products from six bf16 MFMAs per block).  It never enters `value`: the headline stays on the fp32 MFMA.
    alt = None
    if (args.alt_math != 'none' and args.alt_math != args.math and args.math in ('fp32', 'fp32x3') and world == 1
            and graphed is not None and graphed.graph is not None and graphed.failed is None):
        try:                                          # (a failure here must never cost the headline line)
            from iprgan import graphs
            _lib.set_math(args.alt_math)
            g2 = graphs.GraphedStep(model, body, inputs_of(0), warmup=3)
            for i in range(6):                            # three eager steps (autotune of the new tiles), capture, replays
                g2(inputs_of(i))
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for i in range(args.steps):
                g2(inputs_of(i))
            torch.cuda.synchronize()
            ea = time.perf_counter() - ta
            m2 = model.get_metrics()
            assert all(v == v for v in m2.values()), f'non-finite metrics {m2}'
            alt = {'mode': args.alt_math, 'value': round(wl['batch'] * args.steps / ea, 2), 'unit': wl['unit'],
                   'ms_per_step': round(ea / args.steps * 1e3, 3), 'steps': args.steps, 'graph_failed': g2.failed,
                   'note': ('the same model and step on the exact fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 tensors in HBM'
                            if args.alt_math == 'fp32' else
                            'the same model and step with three-plane tensors and six bf16 MFMAs per product block') +
                           '; not part of `value`'}
        except Exception as e:                        # noqa: BLE001
            alt = {'mode': args.alt_math, 'error': f'{type(e).__name__}: {e}'}
        finally:
            _lib.set_math(args.math)
        log(f'alt math {args.alt_math}: {alt}')

    ns = None
    if world == 1 and rank == 0 and args.north_star != 'off' and args.math in ('fp32', 'fp32x3'):
        try:                                          # (a failure here must never cost the headline line)
            ns = north_star_conv(device, [args.math] + (['fp32'] if args.math != 'fp32' else []))
        except Exception as e:                        # noqa: BLE001
            ns = {'error': f'{type(e).__name__}: {e}'}
        log(f'north-star conv: {ns}')

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    # which transport carried the gradient buckets (rccl-abi = iprgan_comm_* of the C ABI; torch.distributed = the
    # fallback / gloo test path), the rank count of the library's communicator, and the time per step the compute stream
    # waited for the exchange (summed over the optimizers' reducers, mean over the last steps)
    from iprgan import parallel
    reducers = {}
    for m in model._modules.values():
        for p in (m.parameters() if isinstance(m, torch.nn.Module) else ()):
            r = parallel.owner_of(p)
            if r is not None:
                reducers[id(r)] = r
    forced = world == 1 and os.environ.get('IPRGAN_FORCE_COMM') == '1'       # one rank through the library's RCCL communicator
    comm = {'transport': parallel.transport_name() if (world > 1 or forced) else 'none', 'nranks': parallel.comm_nranks(),
            'exposed_ms': round(sum(r.exposed_ms() for r in reducers.values()), 4) if (world > 1 or forced) else 0.0}

    if rank == 0:
        B = wl['batch']
        ms = elapsed / args.steps * 1e3
        value = B * world * args.steps / elapsed
        dom = max(kernels, key=lambda k: k['ms']) if kernels else None
        roof = None
        peak_mode = {'fp32': PEAK_FP32_MFMA, 'fp32x3': PEAK_X3_MFMA}.get(args.math, PEAK_BF16_MFMA)
        # profiles/<round>_[<workload>_]pmc_traffic.json; the headline workload has no infix
        tag = '' if args.workload == 'dcgan64' else args.workload + '_'
        if args.math != 'fp32':
            tag = f'{args.workload}_{args.math}_'
        rnd = 'r[0-9][0-9]_'
        traffic = traffic_src = util_pmc = util_src = None
        try:          # HBM bytes per launch from the committed PMC passes (profiles/: separate rocprofv3 runs)
            tf = _latest_profile(f'{rnd}{tag}pmc_traffic.json')
            traffic = json.load(open(tf))['kernels'][dom['name']]['hbm_bytes_per_launch']
            traffic_src = os.path.relpath(tf, ROOT)
        except Exception:
            pass
        try:          # hardware MFMA utilisation of the step's conv kernels from the committed PMC pass
            uf = _latest_profile(f'{rnd}{tag}mfma_util.json')
            util_pmc = json.load(open(uf))['all_conv']['mfma_util_pct']
            util_src = os.path.relpath(uf, ROOT)
        except Exception:
            pass
        if dom:
            ach = dom['flops'] / (dom['ms'] * 1e-3)
            bf16_kernel = args.math not in ('fp32', 'fp32x3') and any(t in dom['name'] for t in ('bf16', 'pipe', 'halo'))
            peak = PEAK_X3_MFMA if 'x3' in dom['name'] else PEAK_BF16_MFMA if bf16_kernel else PEAK_FP32_MFMA
            roof = {'bound': 'mfma', 'kernel': dom['name'], 'achieved': round(ach / 1e12, 2),
                    'peak': round(peak / 1e12, 1), 'unit': 'TFLOP/s',
                    'frac': round(ach / peak, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                    'launches': dom['launches'], 'avg_launch_us': round(dom['ms'] * 1e3 / dom['launches'], 2),
                    'source': f'HIP events on the launch stream, this run: a window of {prof_steps} EAGER (kernel-by-kernel) steps behind '
                              'the closing fence of the timed region - not the graph replays `value` is made of, which cannot carry '
                              'per-kernel events (the committed rocprofv3 --kernel-trace of replays agrees within 2 %: profiles/)'}
            # the three-plane ring exists under three __global__ names (same stages, same six-MFMA product block, same epilogue;
            # four multiplying+loading waves / the 16x16x32 form / dedicated loader waves): the tuner deals a layer to whichever
            # is fastest, so one name alone shows the layers it was dealt, not the algorithm - report the three together as well
            ring = [k for k in kernels if k['name'] in ('gconv_x3p_kernel', 'gconv_x3p16_kernel', 'gconv_x3ws_kernel') and k['ms'] > 0]
            if dom['name'] in [k['name'] for k in ring] and len(ring) > 1:
                rms, rfl = sum(k['ms'] for k in ring), sum(k['flops'] for k in ring)
                roof['ring_family'] = {'kernels': {k['name']: {'launches': k['launches'], 'ms': round(k['ms'], 3),
                                                                 'tflops': round(k['flops'] / (k['ms'] * 1e-3) / 1e12, 2)} for k in ring},
                                       'achieved': round(rfl / (rms * 1e-3) / 1e12, 2), 'frac': round(rfl / (rms * 1e-3) / peak, 4)}
        conv_ms = sum(k['ms'] for k in kernels)
        conv_flops = sum(k['flops'] for k in kernels)
        out = {
            'metric': wl['metric'], 'value': round(value, 2), 'unit': wl['unit'],
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'clock_warm_ms': args.clock_warm_ms, 'ms_per_step': round(ms, 3),
            'ms_per_step_median': round(pct(0.5), 3), 'ms_per_step_p10': round(pct(0.1), 3),
            'ms_per_step_p90': round(pct(0.9), 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if args.math in ('fp32', 'fp32x3') else 'bf16',
            # which arithmetic produced `value` (r03 and earlier: 'fp32' = the exact fp32 MFMA; since r04 the default is
            # 'fp32x3'; the other fp32 mode of the same run is under `alt_math`)
            'math_mode': args.math,
            'data': 'synthetic',
            'config': {'workload': wl['text'] + ', ' +
                                   {'fp32': 'fp32 (exact fp32 MFMA)', 'fp32x3': 'fp32 tensors stored as three exact bf16 planes (x = h + m + l), fp32-grade products from six bf16 MFMAs per block with fp32 accumulation', 'bf16': 'bf16 MFMA tiles (fp32 accumulate, fp32 tensors and master weights)',
                                    'bf16act': 'bf16 MFMA tiles, bf16 activations in HBM (fp32 accumulate, statistics, master weights)'}[args.math] +
                                   ', Adam',
                       'global_batch': B * world, 'parallelism': f'dp{world}'},
            'roofline': roof,
            # the board's shader clock sampled while the timed steps were running; the MFMA peaks above assume 2400 MHz - the
            # six-MFMA kernels run the board at its power cap, below that
            'sclk_mhz_in_timed_region': sclk,
            'mfma_peak_at_that_clock_tflops': round(peak_mode / 1e12 * sclk / 2400.0, 1) if sclk else None,
            'conv_kernels': {'device_ms_per_step': round(conv_ms / prof_steps, 3), 'steps_sampled': prof_steps,
                             'tflops': round(conv_flops / max(conv_ms, 1e-9) / 1e9, 2),
                             'mfma_util_pct': round(100 * conv_flops / max(conv_ms, 1e-9) / 1e-3 / peak_mode, 1),
                             'mfma_util_pct_pmc': util_pmc, 'mfma_util_pct_pmc_source': util_src,
                             'by_kernel': [{'name': k['name'], 'launches': k['launches'],
                                            'ms': round(k['ms'], 2),
                                            'tflops': round(k['flops'] / max(k['ms'], 1e-9) / 1e9, 2)}
                                           for k in kernels]},
            'alt_math': alt,
            'host_enqueue_ms_per_step': round(host_elapsed / args.steps * 1e3, 3),
            # ... of the steps replayed from the captured graph alone (= all timed steps once the step is captured)
            'host_enqueue_ms_per_replayed_step': round(host_replayed / n_replayed * 1e3, 3) if n_replayed else None,
            # what a step costs when it is NOT a graph replay (the eager window behind the timed region; at N > 1 under
            # `--graph auto` the timed region itself is eager): device time per step and host time to enqueue it
            'eager_ms_per_step': round(eager['ms'], 3) if eager else None,
            'host_enqueue_ms_per_eager_step': round(eager['host_ms'], 3) if eager else None,
            'eager_steps_sampled': eager['steps'] if eager else None,
            'graph': ({'policy': args.graph, 'captured': graphed.graph is not None, 'replays_in_timed_region': replays_timed,
                       'eager_steps_in_timed_region': args.steps - replays_timed, 'failed': graphed.failed}
                      if graphed is not None else {'policy': args.graph, 'captured': False, 'replays_in_timed_region': 0,
                                                   'eager_steps_in_timed_region': args.steps, 'failed': None}),
            'transport': comm['transport'], 'comm_nranks': comm['nranks'],
            'allreduce_exposed_ms_per_step': comm['exposed_ms'],
            'step_algorithmic_tflops': round(wl['gflop'] * B * world / ms, 2),
            # (MFMA peak alone: kept for continuity with rounds 1-5; `step_roofline` below is the bound SURVEY 8(d) defines)
            'step_roofline_frac': round(wl['gflop'] * B * world / ms * 1e12 / peak_mode, 4),
            'step_roofline': step_roofline(acct, peak_mode, ms),
            'north_star_conv': ns,
            'metrics_last_step': {k: round(v, 5) for k, v in metrics.items()},
        }
        if baseline is not None:
            out['cpu_baseline'] = baseline
        print(json.dumps(out), flush=True)
    if world > 1:
        parallel.RcclTransport.destroy()
        if not parallel.RcclTransport.abandoned:
            dist.destroy_process_group()
        parallel.finish(0)


if __name__ == '__main__':
    main()
