"""Split-K of three-plane convolutions with few output tiles (gpurun): TFLOP/s fp32-equivalent per forced split count
(1 = the unsplit ring tile) and for the autotuned choice, forward and backward-data.  python scripts/probe/x3_splitk_bench.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402
from x3p_check import timeit  # noqa: E402

dev = torch.device('cuda:0')
LAYERS = [  # name, cin, cout, k, s, p, transposed, H, B
    ('D.conv5 256->256 k4s2 @16 B128', 256, 256, 4, 2, 1, False, 16, 128),
    ('D.conv6 256->512 k3 @8 B128', 256, 512, 3, 1, 1, False, 8, 128),
    ('D.conv4 128->256 k3 @16 B128', 128, 256, 3, 1, 1, False, 16, 128),
    ('G.up0 512->256 T k4s2 @4 B128', 512, 256, 4, 2, 1, True, 4, 128),
    ('VGG 512->512 k3 @6 B64', 512, 512, 3, 1, 1, False, 6, 64),
    ('VGG 512->512 k3 @12 B64', 512, 512, 3, 1, 1, False, 12, 64),
    ('VGG 256->512 k3 @12 B64', 256, 512, 3, 1, 1, False, 12, 64),
    ('D96 512->512 k3s2 @12 B64', 512, 512, 3, 2, 1, False, 12, 64),
    ('VGG 256->256 k3 @24 B64', 256, 256, 3, 1, 1, False, 24, 64),
    ('VGG 128->256 k3 @24 B64', 128, 256, 3, 1, 1, False, 24, 64),
    ('D.conv4 128->256 k3 @16 B256', 128, 256, 3, 1, 1, False, 16, 256),
    ('D.conv6 256->512 k3 @8 B256', 256, 512, 3, 1, 1, False, 8, 256),
]
if len(sys.argv) > 1:
    LAYERS = [l for l in LAYERS if any(a in l[0] for a in sys.argv[1:])]
_lib.set_math('fp32x3')
for name, cin, cout, k, s, p, tr, H, B in LAYERS:
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
    d = spec.desc(B, H, H)
    OH, OW = spec.out_hw(H, H)
    x = ops.to_kind(torch.randn(B, H, H, cin, device=dev), 2)
    dy = ops.to_kind(torch.randn(B, OH, OW, cout, device=dev), 2)
    w = torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), device=dev) * 0.05
    wf, wb = ops.conv_prep(spec, d, w, None, True, True)
    flops = 2.0 * B * (H * H if tr else OH * OW) * cin * cout * k * k
    row = dict(layer=name, gflop=round(flops / 1e9, 2))
    for ks in (1, 2, 3, 4, -1):
        _lib.call('iprgan_debug_force_splitk', ks)
        row['fwd_ks%d' % ks] = round(flops / timeit(lambda: ops.conv_fwd(spec, d, x, wf, None)) / 1e9, 1)
        row['dgrad_ks%d' % ks] = round(flops / timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb)) / 1e9, 1)
    _lib.call('iprgan_debug_force_splitk', -1)
    print(json.dumps(row), flush=True)
