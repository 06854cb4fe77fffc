"""First Adam moments of the generator after step 0 of the 128x128 golden run, bf16act vs the reference fixture."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import torch
from oracle import cases
from iprgan import Config, _lib, models
ref = np.load(os.path.join(ROOT, 'tests/golden/dcgan128_steps_wbox.npz'))
for mode in sys.argv[1:] or ['bf16act']:
    _lib.set_math(mode)
    res = cases.run_dcgan_steps(Config, models, [torch.device('cuda:0')], n_steps=2, batch=8, seed=91, cfg=cases.DCGAN128_CFG, size=128)
    _lib.set_math('fp32')
    print(mode, 'FEWIN', os.environ.get('IPRGAN_FEWIN'))
    for k in ref.files:
        if k.startswith('step0/optG') and k.endswith('exp_avg::asum'):
            a, b = float(res[k]), float(ref[k])
            print(f'  {k:40s} {a:.6g} {b:.6g} rel {abs(a - b) / abs(b):.4f}')
