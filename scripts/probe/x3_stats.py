"""fp32x3: epilogue statistics (conv_fwd stats=True) against the statistics of the stored output (GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
for (B, cin, cout, k, s, p, H) in [(2, 128, 256, 3, 1, 1, 24), (2, 128, 128, 3, 2, 1, 48), (2, 256, 256, 3, 2, 1, 24)]:
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, False)
    d = spec.desc(B, H, H)
    x = torch.randn(B, H, H, cin, device=dev).abs()
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    bias = torch.randn(cout, device=dev)
    for mode in ('fp32', 'fp32x3'):
        for tile in (2, 0, 1):
            _lib.set_math(mode)
            _lib.call('iprgan_debug_force_tiles', tile, -1)
            wf, _ = ops.conv_prep(spec, d, w, None, True, False)
            y, stats = ops.conv_fwd(spec, d, x, wf, bias, stats=True)
            _, mean, invstd = ops.bn_fwd(y, None, None, None, None, 1e-5, 0.0, True, 0, conv_stats=stats, conv_bias=bias)
            y64 = y.double().reshape(-1, y.shape[-1])
            m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
            em = ((mean.double() - m64).abs() / v64.sqrt()).max().item()
            ev = ((invstd.double() - 1 / (v64 + 1e-5).sqrt()).abs() * (v64 + 1e-5).sqrt()).max().item()
            print(f'B{B} {cin}->{cout} k{k}s{s} @{H} {mode:7s} tile {tile}: rows {stats[1]}  |mean - mean(y)|/std {em:.2e}  invstd rel err {ev:.2e}', flush=True)
_lib.call('iprgan_debug_force_tiles', -1, -1)
_lib.set_math('fp32')
