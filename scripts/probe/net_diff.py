"""Which outputs of a network case differ between math modes 'fp32' and 'fp32x3' (gpurun): python scripts/probe/net_diff.py ConvGenerator64"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from iprgan import _lib, networks  # noqa: E402
from oracle import cases  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'ConvGenerator64'
dev = torch.device('cuda:0')
out = {}
for mode in ('fp32', 'fp32x3'):
    _lib.set_math(mode)
    out[mode] = cases.run_net_case(networks, name, device=dev)
bad = 0
for k in sorted(out['fp32']):
    a, b = np.asarray(out['fp32'][k], dtype=np.float64), np.asarray(out['fp32x3'][k], dtype=np.float64)
    if a.dtype.kind not in 'fc' or a.shape != b.shape:
        continue
    d = float(np.abs(a - b).max()) if a.size else 0.0
    s = float(np.abs(a).max()) if a.size else 0.0
    if d > 1e-3 * max(s, 1e-6):
        bad += 1
        print(f'{k}: max diff {d:.3e} of scale {s:.3e}')
print('differing keys:', bad)
