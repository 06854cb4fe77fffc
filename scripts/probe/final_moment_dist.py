"""How far the optimizer moments after the LAST step of the multi-step golden cases sit from the reference's, element by
element (VERDICT r04 next #8b: the magnitude-only 'scale' policy of final/optG, final/optD is replaced by an element-wise
one; this probe measures the distribution the new policy is set from).  GPU box:  python scripts/probe/final_moment_dist.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'ipr-gan_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import cases, gan  # noqa: E402
from iprgan import Config, _lib, models  # noqa: E402

dev = torch.device('cuda:0')
G = lambda n: np.load(os.path.join(ROOT, 'tests', 'golden', n + '.npz'), allow_pickle=False)      # noqa: E731


def report(tag, res, ref, prefixes):
    keys = sorted({k.split('::')[0] for k in (ref.files if hasattr(ref, 'files') else ref) if '::' in k and k.startswith(prefixes)})
    worst, share2, share1, worst_sum = 0.0, 0.0, 0.0, 0.0
    rows = []
    for p in keys:
        big = float(max(np.abs(ref[f'{p}::samp']).max(), np.abs(ref[f'{p}::dense']).max() if f'{p}::dense' in ref else 0.0))
        if big < 1e-5:          # zero-gradient tensors (biases in front of a norm layer): noise on both sides, under the 1e-6 floor
            continue
        w, s2, s1, n = 0.0, 0, 0, 0
        for f in ('head', 'samp', 'dense'):
            if f'{p}::{f}' not in ref:
                continue
            b = np.asarray(ref[f'{p}::{f}'], dtype=np.float64)
            d = np.maximum(0.0, np.abs(np.asarray(res[f'{p}::{f}'], dtype=np.float64) - b) - 2e-2 * np.abs(b) - 1e-6) / big
            w = max(w, float(d.max()))
            s2 += int((d > 2e-2).sum()); s1 += int((d > 1e-2).sum()); n += d.size
        rs = max(abs(float(res[f'{p}::{f}']) - float(ref[f'{p}::{f}'])) / max(1e-30, abs(float(ref[f'{p}::{f}']))) for f in ('asum', 'l2') if f'{p}::{f}' in ref)
        rows.append((w, s2 / n, s1 / n, rs, p))
        worst, share2, share1, worst_sum = max(worst, w), max(share2, s2 / n), max(share1, s1 / n), max(worst_sum, rs)
    rows.sort(reverse=True)
    print(f'{tag}: {len(rows)} tensors; worst |d|/scale {worst:.4f}; max share > 2 % {share2:.4f}, > 1 % {share1:.4f}; worst asum / l2 rel {worst_sum:.4f}')
    for r in rows[:4]:
        print(f'    {r[4]:60s} worst {r[0]:.4f} share>2% {r[1]:.4f} share>1% {r[2]:.4f} sums {r[3]:.4f}')
    sys.stdout.flush()


for mode in ('fp32', 'fp32x3'):
    _lib.set_math(mode)
    fin = ('final/optG', 'final/optD')
    for wbox in (True, False):
        steps = 3 if wbox else 2
        report(f'[{mode}] dcgan wbox={wbox}', cases.run_dcgan_steps(Config, models, [dev], n_steps=steps, wbox=wbox),
               G('dcgan_steps_wbox' if wbox else 'dcgan_steps_plain'), fin)
    report(f'[{mode}] srgan', cases.run_srgan_steps(Config, models, [dev]), G('srgan_steps_wbox'), fin)
    report(f'[{mode}] cyclegan', cases.run_cyclegan_steps(Config, models, [dev]), G('cyclegan_steps_wbox'), fin)
    report(f'[{mode}] vae', cases.run_vae_steps(Config, models, [dev]), G('vae_steps_wbox'), ('final/opt',))
    report(f'[{mode}] dcgan complete', cases.run_dcgan_complete_steps(Config, models, [dev]), G('dcgan_steps_complete'), fin)
    for name, fn in (('dcgan128', None),):
        pass
    # the live-oracle CycleGAN 'complete' case: step-0 moments (the second 'scale' policy)
    import tempfile
    from PIL import Image
    rgba = np.zeros((40, 40, 4), dtype=np.uint8)
    rgba[6:34, 10:30] = (220, 40, 90, 255)
    rgba[14:22, 14:26] = (20, 200, 120, 255)
    path = os.path.join(tempfile.mkdtemp(), 'logo.png')
    Image.fromarray(rgba, 'RGBA').save(path)
    bb = cases.bbox_cfg('translation', path, 16, 16)
    res = cases.run_cyclegan_steps(Config, models, [dev], n_steps=1, bbox=bb)
    ref = cases.run_cyclegan_steps(gan.Cfg, gan, gan.CPU, n_steps=1, bbox=bb)
    report(f'[{mode}] cyclegan complete step0', res, ref, ('step0/optD', 'step0/optG'))
