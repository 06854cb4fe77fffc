"""Backward-weight of three-plane tensors per candidate on the DCGAN-64 / north-star shapes (TFLOP/s fp32-equivalent)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
LAYERS = [('D.conv1 64->64 k4s2', 64, 64, 4, 2, 1, False, 64, 128), ('D.conv2 64->128 k3', 64, 128, 3, 1, 1, False, 32, 128),
          ('D.conv3 128->128 k4s2', 128, 128, 4, 2, 1, False, 32, 128), ('D.conv4 128->256 k3', 128, 256, 3, 1, 1, False, 16, 128),
          ('D.conv5 256->256 k4s2', 256, 256, 4, 2, 1, False, 16, 128), ('D.conv6 256->512 k3', 256, 512, 3, 1, 1, False, 8, 128),
          ('G.up0 512->256 T', 512, 256, 4, 2, 1, True, 8, 128), ('G.up1 256->128 T', 256, 128, 4, 2, 1, True, 16, 128),
          ('G.up2 128->64 T', 128, 64, 4, 2, 1, True, 32, 128), ('NS 256->256 k3 @64 B64', 256, 256, 3, 1, 1, False, 64, 64),
          ('SR 64->64 k3 @24 B64', 64, 64, 3, 1, 1, False, 24, 64)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


_lib.set_math('fp32x3')
for name, cin, cout, k, s, p, tr, H, B in LAYERS:
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
    d = spec.desc(B, H, H)
    OH, OW = spec.out_hw(H, H)
    x = ops.to_kind(torch.randn(B, H, H, cin, device=dev), 2)
    dy = ops.to_kind(torch.randn(B, OH, OW, cout, device=dev), 2)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    flops = 2.0 * B * (H * H if tr else OH * OW) * cin * cout * k * k
    row = dict(layer=name)
    for cand in [int(c) for c in os.environ.get("WX3_CANDS", "-1,0,3,74,75,76").split(",")]:
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        t = timeit(lambda: ops.conv_bwd_weight(spec, d, x, dy, wshape, False))
        row['c%d' % cand] = round(flops / t / 1e9, 1)
    print(json.dumps(row), flush=True)
