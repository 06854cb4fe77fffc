"""Per-tile overhead of a ring tile: time per tile = a + b * (K steps), fitted over layers that differ only in K
(k1 / k3 kernels on the same rows), with and without the epilogue (IPRGAN_X3WS_PROBE=256, M16 loader-wave tiles only).
gpurun: python scripts/probe/tile_overhead.py <tile>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 36
BM, BN = {36: (128, 128), 35: (256, 64), 37: (128, 64), 33: (128, 128), 19: (128, 128), 29: (128, 128)}.get(tile, (128, 128))
_lib.set_math('fp32x3')


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3      # us


rows = []
for (cin, k) in [(32, 1), (64, 1), (128, 1), (256, 1), (64, 3), (128, 3), (256, 3)]:
    for (B, H) in [(64, 64), (8, 64)]:
        cout = BN
        spec = ops.ConvSpec(cin, cout, k, 1, k // 2, 0, False)
        d = spec.desc(B, H, H)
        x = ops.to_kind(torch.randn(B, H, H, cin, device=dev), 2)
        w = torch.randn(cout, cin, k, k, device=dev) * 0.05
        wf, _ = ops.conv_prep(spec, d, w, None, True, False)
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        t = timeit(lambda: ops.conv_fwd(spec, d, x, wf, None))
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        M = B * H * H
        tiles = (M // BM) * (cout // BN)
        steps = cin * k * k // 32
        per_cu = max(1.0, tiles / 256)
        rows.append(dict(cin=cin, k=k, B=B, steps=steps, tiles=tiles, us=round(t, 1), us_per_tile=round(t / per_cu, 2)))
        print(json.dumps(rows[-1]), flush=True)
for B in (64, 8):
    sel = [r for r in rows if r['B'] == B]
    n = len(sel)
    sx = sum(r['steps'] for r in sel); sy = sum(r['us_per_tile'] for r in sel)
    sxx = sum(r['steps'] ** 2 for r in sel); sxy = sum(r['steps'] * r['us_per_tile'] for r in sel)
    b = (n * sxy - sx * sy) / (n * sxx - sx * sx)
    a = (sy - b * sx) / n
    print(json.dumps(dict(tile=tile, B=B, probe=os.environ.get('IPRGAN_X3WS_PROBE', '0'), a_us_per_tile=round(a, 2), b_us_per_step=round(b, 3))), flush=True)
