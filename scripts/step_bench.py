"""Secondary configs of BASELINE.json (not the headline metric): SRGAN GAN-phase step (24->96, batch 64,
VGG19 features with random weights), CycleGAN step (Resnet9Blocks + ConvDiscriminator, 256x256, batch 8) and the
DCGAN step in bf16 math (64x64 batch 128, and BASELINE config 5: 128x128 batch 256), all with the sign-loss
wrapper, on one GPU.  Prints one JSON line per config."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import Config, models  # noqa: E402

dev = torch.device('cuda:0')
WBOX = {'gamma_0': 0.1, 'string': 'EXAMPLE A'}


def timed(fn, warm, steps):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def srgan(B=64):
    m = models.SRGAN(Config({'G': 'SRResNet', 'D': 'Discriminator96', 'V': 'VGG19Feature', 'opt': 'Adam',
                             'opt_param': {'lr': 1e-4}}), device=[dev])
    m = models.WhiteBoxWrapper(m, Config(dict(WBOX, target='G')))
    lr, hr = torch.rand(B, 3, 24, 24, device=dev), torch.rand(B, 3, 96, 96, device=dev)

    def step():
        m.update_g({'low_res': lr, 'high_res': hr, 'pretrain': False})
        m.update_d({'high_res': m.high_res, 'super_res': m.super_res})
    dt = timed(step, 3, 10)
    print(json.dumps({'config': f'SRGAN 24->96 GAN phase B={B} fp32', 'ms_per_step': round(dt * 1e3, 2),
                      'img_per_s': round(B / dt, 1), 'algorithmic_tflops': round(43.31 * B / dt / 1e3, 1)}), flush=True)


def cyclegan(B=8, S=256):
    m = models.CycleGAN(Config({'G': 'Resnet9Blocks', 'D': 'ConvDiscriminator', 'opt': 'Adam',
                                'opt_param': {'lr': 2e-4, 'betas': [0.5, 0.999]}, 'pool_size': 50, 'lambda_A': 10.0,
                                'lambda_B': 10.0, 'lambda_idt': 0.5, 'epoch': 200}), device=[dev])
    m = models.WhiteBoxWrapper(m, Config(dict(WBOX, target='GB')))
    a, b = torch.tanh(torch.randn(B, 3, S, S, device=dev)), torch.tanh(torch.randn(B, 3, S, S, device=dev))

    def step():
        m.update_g({'real_A': a, 'real_B': b})
        m.update_d({'real_A': m.real_A, 'real_B': m.real_B, 'fake_A': m.fake_A.detach(), 'fake_B': m.fake_B.detach()})
    dt = timed(step, 2, 5)
    print(json.dumps({'config': f'CycleGAN Resnet9 {S}x{S} B={B} fp32', 'ms_per_step': round(dt * 1e3, 1),
                      'pairs_per_s': round(B / dt, 2), 'algorithmic_tflops': round(1884.6 * B / dt / 1e3, 1)}), flush=True)


def dcgan(size, B, math):
    from iprgan import _lib
    _lib.set_math(math)
    try:
        m = models.DCGAN(Config({'G': f'ConvGenerator{size}', 'D': f'SNDiscriminator{size}', 'opt': 'Adam',
                                 'opt_param': {'lr': 2e-4, 'betas': [0.5, 0.999]}}), device=[dev])
        m = models.WhiteBoxWrapper(m, Config(dict(WBOX, target='G')))
        x, z = torch.tanh(torch.randn(B, 3, size, size, device=dev)), torch.randn(B, 128, device=dev)

        def step():
            m.update_d({'real_sample': x, 'latent': z})
            m.update_g({'fake_sample': m.fake_sample})
        dt = timed(step, 6, 20)
        gflop_img = 9.443 if size == 64 else 37.77            # 3 F_G + 8 F_D
        metrics = m.get_metrics()
        print(json.dumps({'config': f'DCGAN-{size} + sign loss B={B} {math} math (fp32 tensors / master weights)',
                          'ms_per_step': round(dt * 1e3, 2), 'img_per_s': round(B / dt, 1),
                          'algorithmic_tflops': round(gflop_img * B / dt / 1e3, 1),
                          'finite': all(v == v and abs(v) < 1e9 for v in metrics.values()),
                          'ber': float(m.loss_model.compute_ber(m.G))}), flush=True)
    finally:
        _lib.set_math('fp32')


if __name__ == '__main__':
    which = sys.argv[1:] or ['srgan', 'cyclegan', 'dcgan']
    if 'dcgan' in which:
        dcgan(64, 128, 'bf16')
        dcgan(128, 256, 'fp32')
        dcgan(128, 256, 'bf16')
    if 'srgan' in which:
        srgan()
    if 'cyclegan' in which:
        cyclegan()
