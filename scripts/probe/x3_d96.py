"""Discriminator96 (batch 2) in fp32x3 against float64, per forced gconv tile (GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch  # noqa: E402
import test_gpu_models as T  # noqa: E402
from iprgan import networks, _lib  # noqa: E402

dev = torch.device('cuda:0')
attr, kw, xshape, seed = T.cases.NET_CASES['Discriminator96']
XS = int(os.environ.get('X3_XSEED', '1000'))
x = torch.tanh(T.recipe.tensor(seed, XS, xshape))


def make(mod):
    net = getattr(mod, attr)(**kw)
    T.recipe.fill(net, seed)
    return net


t64 = T._net_pass(make(T.nets), x, seed, torch.float64, 'cpu')
keys = ['out', 'dx', 'grad/4.1.bias', 'grad/4.1.weight', 'grad/5.1.bias', 'grad/2.1.bias']
for mode in ('fp32', 'fp32x3'):
    for tile in [int(t) for t in os.environ.get('X3_TILES', '-1').split(',')]:
        for wg in (-1,):
            _lib.set_math(mode)
            _lib.call('iprgan_debug_force_tiles', tile, wg)
            eng = T._net_pass(make(networks), x, seed, torch.float32, dev)
            errs = ' '.join('%s %.1e' % (k, float((eng[k] - t64[k]).pow(2).mean().sqrt()) / float(t64[k].abs().max())) for k in keys)
            print(f'{mode:7s} tile {tile:2d}: {errs}', flush=True)
_lib.call('iprgan_debug_force_tiles', -1, -1)
_lib.set_math('fp32')
