"""Times every backward-weight candidate (tile shape x split target x LDS-image variant) on selected layers.
usage: python scripts/wgrad_sweep.py [layer substring]   (GPU only)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch  # noqa: E402
from iprgan import _lib, ops  # noqa: E402
from conv_bench import LAYERS, timeit, B  # noqa: E402

SHAPES = ['128x128', '64x64', '128x64', '128x128,8w']
TARGETS = [768, 1536, 3072, 6144, 384]


def main():
    dev = torch.device('cuda:0')
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, cin, cout, k, s, p, tr, H in LAYERS:
        if only and only not in name:
            continue
        b = 64 if name.startswith('NS') else B
        spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=1 if 'reflect' in name else 0)
        d = spec.desc(b, H, H)
        OH, OW = spec.out_hw(H, H)
        x = torch.randn(b, H, H, ops.c4(cin), device=dev)
        dy = torch.randn(b, OH, OW, ops.c4(cout), device=dev)
        wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        flops = 2.0 * b * (H * H if tr else OH * OW) * cin * cout * k * k
        res = {}
        for cand in list(range(60)) + [71, 72, 73]:
            _lib.call('iprgan_debug_force_tiles', -1, cand)
            t = timeit(lambda: ops.conv_bwd_weight(spec, d, x, dy, wshape, False), n=5)
            res[cand] = t
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        best = sorted(res, key=res.get)[:6]
        row = {'layer': name, 'gflop': round(flops / 1e9, 1)}
        for v in range(3):
            vb = min(range(20 * v, 20 * v + 20), key=res.get)
            row[f'v{v}'] = f'{SHAPES[vb % 4]}/{TARGETS[(vb % 20) // 4]}: {res[vb] * 1e3:.1f}us {flops / res[vb] / 1e9:.1f}TF'
        row['halo32'] = {c: f'{res[c] * 1e3:.1f}us {flops / res[c] / 1e9:.1f}TF' for c in (71, 72, 73)}
        row['top'] = [f'{c}:{res[c] * 1e3:.1f}' for c in best]
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
