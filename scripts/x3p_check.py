"""Three-plane ('fp32x3' with IPRGAN_ST_X3 tensors) convolution tiles on the GPU: accuracy against float64 on small layers,
then TFLOP/s (fp32-equivalent) per tile on the DCGAN-64 / north-star shapes.  Usage: python scripts/x3p_check.py [acc|bench|all]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
SMALL = [  # B, cin, cout, k, stride, pad, H, transposed, reflect
    (8, 64, 128, 3, 1, 1, 32, False, False), (8, 128, 128, 4, 2, 1, 32, False, False), (4, 256, 512, 3, 1, 1, 8, False, False),
    (8, 64, 64, 4, 2, 1, 64, False, False), (8, 256, 128, 4, 2, 1, 16, True, False), (2, 256, 256, 3, 2, 1, 24, False, False),
    (4, 96, 160, 3, 1, 1, 20, False, False), (2, 128, 128, 3, 1, 1, 16, False, True), (3, 64, 192, 3, 1, 1, 20, False, False),
]
TILES = [-1, 0, 18, 26, 27, 28, 29, 30, 31, 32, 33, 36, 37]


def relerr(got, want):
    return float((got.double().cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())


def acc():
    bad = 0
    for (B, cin, cout, k, s, p, H, tr, refl) in SMALL:
        g = torch.Generator().manual_seed(1234 + cin + cout)
        spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=1 if refl else 0)
        OH, OW = spec.out_hw(H, H)
        x = (torch.randn(B, H, H, cin, generator=g) + 0.5).to(dev)
        dy = torch.randn(B, OH, OW, cout, generator=g).to(dev)
        w = (torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), generator=g) * 0.05).to(dev)
        bias = torch.randn(cout, generator=g).to(dev)
        x64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
        w64 = w.double().cpu().requires_grad_(True)
        xin = F.pad(x64, (p, p, p, p), mode='reflect') if refl else x64
        y64 = F.conv_transpose2d(xin, w64, bias.double().cpu(), s, p) if tr else F.conv2d(xin, w64, bias.double().cpu(), s, 0 if refl else p)
        y64.backward(dy.double().cpu().permute(0, 3, 1, 2))
        ref = (y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1))
        _lib.set_math('fp32')
        d0 = spec.desc(B, H, H)
        wf0, wb0 = ops.conv_prep(spec, d0, w, None, True, True)
        e32 = (relerr(ops.conv_fwd(spec, d0, x, wf0, bias), ref[0]),
               relerr(ops.conv_bwd_data(spec, d0, dy, wb0), ref[1]) if not refl else 0.0)
        _lib.set_math('fp32x3')
        assert _lib.act_x3()
        d = spec.desc(B, H, H)
        assert d.x_bf16 == 2 and d.y_bf16 == 2
        wf, wb = ops.conv_prep(spec, d, w, None, True, True)
        xp, dyp = ops.to_kind(x, 2), ops.to_kind(dy, 2)
        assert float((ops.to_kind(xp, 0) - x).abs().max()) == 0.0, 'split / join must be exact'
        for tile in TILES:
            _lib.call('iprgan_debug_force_tiles', tile, -1)
            y = ops.conv_fwd(spec, d, xp, wf, bias)
            ey = relerr(ops.to_kind(y, 0), ref[0])
            edx = 0.0
            if not refl:
                dx = ops.conv_bwd_data(spec, d, dyp, wb)
                edx = relerr(ops.to_kind(dx, 0), ref[1])
            ok = ey <= 1.25 * e32[0] + 1e-7 and edx <= 1.25 * e32[1] + 1e-7
            bad += 0 if ok else 1
            print(json.dumps(dict(layer=[B, cin, cout, k, s, p, H, tr, refl], tile=tile, y=ey, dx=edx, y32=e32[0], dx32=e32[1], ok=ok)), flush=True)
        # statistics + fused derivative + residual through the three-plane epilogue (tile 18 and the register-staged tile 0)
        if not tr and not refl and s == 1:
            for tile in (18, 26, 28, 30, 32, 33, 36, 0):
                _lib.call('iprgan_debug_force_tiles', tile, -1)
                y, stats = ops.conv_fwd(spec, d, xp, wf, bias, stats=True)
                yf = ops.to_kind(y, 0)
                _, mean, invstd = ops.bn_fwd(yf, None, None, None, None, 1e-5, 0.0, True, 0, conv_stats=stats, conv_bias=bias)
                y2 = yf.double().reshape(-1, yf.shape[-1])
                m64, v64 = y2.mean(0), y2.var(0, unbiased=False)
                es = float(((mean.double() - m64).abs() / v64.sqrt()).max())
                prev = ops.to_kind(torch.randn(B, H, H, cin, generator=g).to(dev), 2)
                res = ops.to_kind(torch.randn(B, H, H, cin, generator=g).to(dev), 2)
                dx = ops.conv_bwd_data(spec, d, dyp, wb, prev, _lib.ACT_LRELU, 0.2, residual=res)
                want = ref[1] * torch.where(ops.to_kind(prev, 0).double().cpu() > 0, 1.0, 0.2) + ops.to_kind(res, 0).double().cpu()
                ef = relerr(ops.to_kind(dx, 0), want)
                ok = es < 2e-6 and ef < 1e-6
                bad += 0 if ok else 1
                print(json.dumps(dict(layer=[B, cin, cout, k, s, p, H], tile=tile, stats_err=es, fused_err=ef, ok=ok)), flush=True)
        _lib.call('iprgan_debug_force_tiles', -1, -1)
    _lib.set_math('fp32')
    print(json.dumps(dict(acc_failures=bad)), flush=True)
    return bad


BENCH = [  # name, cin, cout, k, s, p, transposed, H, B, reflect
    ('NS 256->256 k3 @64 B64', 256, 256, 3, 1, 1, False, 64, 64, False),
    ('NS(iii) reflect', 256, 256, 3, 1, 1, False, 64, 64, True),
    ('D.conv1 64->64 k4s2', 64, 64, 4, 2, 1, False, 64, 128, False),
    ('D.conv2 64->128 k3', 64, 128, 3, 1, 1, False, 32, 128, False),
    ('D.conv3 128->128 k4s2', 128, 128, 4, 2, 1, False, 32, 128, False),
    ('D.conv4 128->256 k3', 128, 256, 3, 1, 1, False, 16, 128, False),
    ('D.conv5 256->256 k4s2', 256, 256, 4, 2, 1, False, 16, 128, False),
    ('D.conv6 256->512 k3', 256, 512, 3, 1, 1, False, 8, 128, False),
    ('G.up0 512->256 T k4s2', 512, 256, 4, 2, 1, True, 8, 128, False),
    ('G.up1 256->128 T k4s2', 256, 128, 4, 2, 1, True, 16, 128, False),
    ('G.up2 128->64 T k4s2', 128, 64, 4, 2, 1, True, 32, 128, False),
    ('SR 64->64 k3 @24 B64', 64, 64, 3, 1, 1, False, 24, 64, False),
]


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


def bench():
    _lib.set_math('fp32x3')
    tiles = [int(t) for t in os.environ.get('X3P_TILES', '-1,18,21,26,27').split(',')]
    for name, cin, cout, k, s, p, tr, H, B, refl in BENCH:
        spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=1 if refl else 0)
        d = spec.desc(B, H, H)
        OH, OW = spec.out_hw(H, H)
        x = ops.to_kind(torch.randn(B, H, H, cin, device=dev), 2)
        dy = ops.to_kind(torch.randn(B, OH, OW, cout, device=dev), 2)
        w = torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), device=dev) * 0.05
        wf, wb = ops.conv_prep(spec, d, w, None, True, True)
        flops = 2.0 * B * (H * H if tr else OH * OW) * cin * cout * k * k
        row = dict(layer=name, gflop=round(flops / 1e9, 2))
        for tile in tiles:
            _lib.call('iprgan_debug_force_tiles', tile, -1)
            t_f = timeit(lambda: ops.conv_fwd(spec, d, x, wf, None))
            row['fwd_t%d' % tile] = round(flops / t_f / 1e9, 1)
            if not refl:
                t_d = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb))
                row['dgrad_t%d' % tile] = round(flops / t_d / 1e9, 1)
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        print(json.dumps(row), flush=True)
    _lib.set_math('fp32')


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'all'
    rc = 0
    if what in ('acc', 'all'):
        rc = acc()
    if what in ('bench', 'all'):
        bench()
    sys.exit(1 if rc else 0)
