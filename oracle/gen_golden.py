"""Generate tests/golden/*.npz by running the REAL reference from /root/reference.

TEST INFRASTRUCTURE, build-container only (the reference is absent on the GPU box).
Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden [fixture names ...]   (default: all)
Only inputs-by-recipe, outputs and summaries are stored - never reference source.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
from . import cases, ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


ONLY = set(sys.argv[1:])          # optional: names of the fixtures to (re)generate; default all


def save(name, res):
    if ONLY and name not in ONLY:
        return
    if callable(res):
        res = res()
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **res)
    print(f'{name}: {len(res)} arrays, {os.path.getsize(path) / 1024:.1f} KiB')


def main():
    assert ref_loader.available(), 'reference not mounted'
    torch.manual_seed(0)
    torch.set_num_threads(8)
    networks, tools, models = ref_loader.load()
    os.makedirs(OUT, exist_ok=True)
    for name in cases.NET_CASES:
        save('net_' + name, lambda: cases.run_net_case(networks, name))
    for name in cases.SIGN_CASES:
        save('sign_' + name, lambda: cases.run_sign_case(networks, tools.SignLossModel, ref_loader.Config, name))
    # known-answer vector from SURVEY section 4: first 16 signs of 'EXAMPLE A'
    bg = tools.BitGenerator('EXAMPLE A')
    save('bits_EXAMPLE_A', lambda: {'bits': np.array(bg.get(200), dtype=np.int8)})
    dev = [torch.device('cpu')]
    save('dcgan_steps_wbox', lambda: cases.run_dcgan_steps(ref_loader.Config, models, dev, wbox=True))
    save('dcgan_steps_plain', lambda: cases.run_dcgan_steps(ref_loader.Config, models, dev, n_steps=2, wbox=False))
    save('srgan_steps_wbox', lambda: cases.run_srgan_steps(ref_loader.Config, models, dev))
    save('cyclegan_steps_wbox', lambda: cases.run_cyclegan_steps(ref_loader.Config, models, dev))
    save('vae_steps_wbox', lambda: cases.run_vae_steps(ref_loader.Config, models, dev))
    save('bbox_transforms', lambda: cases.run_bbox_transforms(tools, ref_loader.Config))
    save('dcgan128_steps_wbox', lambda: cases.run_dcgan_steps(ref_loader.Config, models, dev, n_steps=2, batch=8, seed=91,
                                                      cfg=cases.DCGAN128_CFG, size=128))
    save('cyclegan_pool_steps', lambda: cases.run_cyclegan_pool_steps(ref_loader.Config, models, dev))
    save('vgg_layer_names', lambda: {'names': np.array(cases.vgg_layer_names_from_source(
        os.path.join(ref_loader.REF, 'networks', 'vgg.py')))})
    save('loss_factories', lambda: cases.run_loss_factories(ref_loader.load_loss()))
    save('dcgan_steps_complete', lambda: cases.run_dcgan_complete_steps(ref_loader.Config, models, dev))


if __name__ == '__main__':
    main()
