"""Per-layer conv microbenchmark (GPU only): times forward / backward-data / backward-weight of every
conv layer of DCGAN-64 at batch 128 (and the north-star 3x3 256->256 @64x64 B=64 shape) through the
C ABI and prints algorithmic TFLOP/s against the 157.3 TFLOP/s fp32 MFMA peak."""
import os
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import ops  # noqa: E402

B = int(os.environ.get("CONV_BENCH_B", "128"))
LAYERS = [  # name, cin, cout, k, s, p, transposed, H
    ('D.conv0 3->64 k3', 3, 64, 3, 1, 1, False, 64),
    ('D.conv1 64->64 k4s2', 64, 64, 4, 2, 1, False, 64),
    ('D.conv2 64->128 k3', 64, 128, 3, 1, 1, False, 32),
    ('D.conv3 128->128 k4s2', 128, 128, 4, 2, 1, False, 32),
    ('D.conv4 128->256 k3', 128, 256, 3, 1, 1, False, 16),
    ('D.conv5 256->256 k4s2', 256, 256, 4, 2, 1, False, 16),
    ('D.conv6 256->512 k3', 256, 512, 3, 1, 1, False, 8),
    ('G.fc 128->32768', 128, 32768, 1, 1, 0, False, 1),
    ('G.up0 512->256 T k4s2', 512, 256, 4, 2, 1, True, 8),
    ('G.up1 256->128 T k4s2', 256, 128, 4, 2, 1, True, 16),
    ('G.up2 128->64 T k4s2', 128, 64, 4, 2, 1, True, 32),
    ('G.out 64->3 T k3', 64, 3, 3, 1, 1, True, 64),
    ('NS 256->256 k3 @64 B64', 256, 256, 3, 1, 1, False, 64),
    # SURVEY section 8(d) microbench: the 3x3 convs Resnet9Blocks executes on a 64x3x256x256 batch
    ('NS(i) 64->128 k3s2 @256 B64', 64, 128, 3, 2, 1, False, 256),
    ('NS(ii) 128->256 k3s2 @128 B64', 128, 256, 3, 2, 1, False, 128),
    ('NS(iii) 256->256 k3 reflect @64 B64', 256, 256, 3, 1, 1, False, 64),
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


def main():
    dev = torch.device('cuda:0')
    from iprgan import _lib
    _lib.set_math(os.environ.get('CONV_BENCH_MATH', 'fp32'))
    if os.environ.get('CONV_BENCH_TILE'):            # force one forward / backward-data tile (include/iprgan.h: iprgan_debug_force_tiles)
        _lib.call('iprgan_debug_force_tiles', int(os.environ['CONV_BENCH_TILE']), -1)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    rows = []
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
        w = torch.randn(*wshape, device=dev) * 0.05
        wf, wb = ops.conv_prep(spec, d, w, None, True, True)
        del w
        flops = 2.0 * b * (H * H if tr else OH * OW) * cin * cout * k * k
        t_f = timeit(lambda: ops.conv_fwd(spec, d, x, wf, None))
        t_d = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb))
        t_w = timeit(lambda: ops.conv_bwd_weight(spec, d, x, dy, wshape, False))
        row = dict(layer=name, gflop=round(flops / 1e9, 2),
                   fwd_us=round(t_f * 1e3, 1), fwd_tf=round(flops / t_f / 1e9, 1),
                   dgrad_us=round(t_d * 1e3, 1), dgrad_tf=round(flops / t_d / 1e9, 1),
                   wgrad_us=round(t_w * 1e3, 1), wgrad_tf=round(flops / t_w / 1e9, 1))
        rows.append(row)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
