"""Generate tests/golden/*.npz by running the REAL reference from /root/reference.

TEST INFRASTRUCTURE, build-container only (the reference is absent on the GPU box).
Run:  PYTHONDONTWRITEBYTECODE=1 python -m oracle.gen_golden
Only inputs-by-recipe, outputs and summaries are stored - never reference source.
"""
import os
import sys

import numpy as np
import torch

sys.dont_write_bytecode = True
from . import cases, ref_loader  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')


def save(name, res):
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
        save('net_' + name, cases.run_net_case(networks, name))
    for name in cases.SIGN_CASES:
        save('sign_' + name, cases.run_sign_case(networks, tools.SignLossModel, ref_loader.Config, name))
    # known-answer vector from SURVEY section 4: first 16 signs of 'EXAMPLE A'
    bg = tools.BitGenerator('EXAMPLE A')
    save('bits_EXAMPLE_A', {'bits': np.array(bg.get(200), dtype=np.int8)})
    dev = [torch.device('cpu')]
    save('dcgan_steps_wbox', cases.run_dcgan_steps(ref_loader.Config, models, dev, wbox=True))
    save('dcgan_steps_plain', cases.run_dcgan_steps(ref_loader.Config, models, dev, n_steps=2, wbox=False))
    save('srgan_steps_wbox', cases.run_srgan_steps(ref_loader.Config, models, dev))
    save('cyclegan_steps_wbox', cases.run_cyclegan_steps(ref_loader.Config, models, dev))
    save('vae_steps_wbox', cases.run_vae_steps(ref_loader.Config, models, dev))
    save('bbox_transforms', cases.run_bbox_transforms(tools, ref_loader.Config))
    save('dcgan_steps_complete', cases.run_dcgan_complete_steps(ref_loader.Config, models, dev))


if __name__ == '__main__':
    main()
