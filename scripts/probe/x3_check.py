"""fp32x3 math mode: error against a float64 convolution, next to the fp32 mode's error (GPU only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
CASES = [(8, 64, 128, 3, 1, 1, 32), (8, 128, 128, 4, 2, 1, 32), (4, 256, 512, 3, 1, 1, 8), (8, 64, 64, 4, 2, 1, 64)]
if os.environ.get('X3_CASES') == 'd96':       # Discriminator96 at batch 2 (the golden fixture's size)
    CASES = [(2, 64, 64, 3, 2, 1, 96), (2, 64, 128, 3, 1, 1, 48), (2, 128, 128, 3, 2, 1, 48), (2, 128, 256, 3, 1, 1, 24),
             (2, 256, 256, 3, 2, 1, 24), (2, 256, 512, 3, 1, 1, 12), (2, 512, 512, 3, 2, 1, 12), (2, 512, 1024, 6, 1, 0, 6),
             (2, 1024, 1, 1, 1, 0, 1)]
for (B, cin, cout, k, s, p, H) in CASES:
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, False)
    d = spec.desc(B, H, H)
    OH, OW = spec.out_hw(H, H)
    x = torch.randn(B, H, H, cin, device=dev) + 0.5
    dy = torch.randn(B, OH, OW, cout, device=dev)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    x64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().cpu().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, s, p)
    y64.backward(dy.double().cpu().permute(0, 3, 1, 2))
    ref = (y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1), w64.grad)
    for tile in ([-1] if len(sys.argv) < 2 else [int(t) for t in sys.argv[1].split(',')]):
        for mode in ('fp32', 'fp32x3'):
            _lib.set_math(mode)
            _lib.call('iprgan_debug_force_tiles', tile, -1)
            wf, wb = ops.conv_prep(spec, d, w, None, True, True)
            y = ops.conv_fwd(spec, d, x, wf, None)
            dx = ops.conv_bwd_data(spec, d, dy, wb)
            dw = ops.conv_bwd_weight(spec, d, x, dy, tuple(w.shape), False)
            if isinstance(dw, tuple):
                dw = dw[0]
            errs = []
            for got, want in zip((y, dx, dw), ref):
                e = (got.double().cpu() - want)
                errs.append('%.2e/%.2e' % (e.abs().max().item() / want.abs().max().item(), (e.pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()))
            print(f'B{B} {cin}->{cout} k{k}s{s} @{H} tile {tile:2d} {mode:7s} max/rms rel err  fwd {errs[0]}  dgrad {errs[1]}  wgrad {errs[2]}', flush=True)
_lib.set_math('fp32')
