"""Per-layer, per-tile conv microbenchmark for BASELINE config 5 (DCGAN-128, batch 256, bf16 activations in HBM; GPU
only): forward and backward-data of every MFMA-path layer of ConvGenerator(mg=16) / SNDiscriminator(md=16) under each
forced tile (0-7: register-staged tiles of conv_igemm.hip, 8-11: LDS-DMA ring tiles of conv_pipe.hip, -1: autotuned)
and backward-weight (autotuned), as algorithmic TFLOP/s against the 2.5 PFLOP/s dense bf16 peak."""
import os
import sys
import json

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

B = int(os.environ.get("CONV_BENCH_B", "256"))
LAYERS = [  # name, cin, cout, k, s, p, transposed, H
    ('D.conv0 3->64 k3', 3, 64, 3, 1, 1, False, 128),
    ('D.conv1 64->64 k4s2', 64, 64, 4, 2, 1, False, 128),
    ('D.conv2 64->128 k3', 64, 128, 3, 1, 1, False, 64),
    ('D.conv3 128->128 k4s2', 128, 128, 4, 2, 1, False, 64),
    ('D.conv4 128->256 k3', 128, 256, 3, 1, 1, False, 32),
    ('D.conv5 256->256 k4s2', 256, 256, 4, 2, 1, False, 32),
    ('D.conv6 256->512 k3', 256, 512, 3, 1, 1, False, 16),
    ('G.up0 512->256 T k4s2', 512, 256, 4, 2, 1, True, 16),
    ('G.up1 256->128 T k4s2', 256, 128, 4, 2, 1, True, 32),
    ('G.up2 128->64 T k4s2', 128, 64, 4, 2, 1, True, 64),
    ('GEMM 1024->1024 k1 @32', 1024, 1024, 1, 1, 0, False, 32),
    ('GEMMsmall 1024->4096 k1 @8', 1024, 4096, 1, 1, 0, False, 8),      # both operands stay in the Infinity Cache
    ('GEMMdeep 4096->1024 k1 @8', 4096, 1024, 1, 1, 0, False, 8),     # plain GEMM: the ring's inner loop without the im2col walk
]


def timeit(fn, n=6):
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
    _lib.set_math('bf16act')
    only = sys.argv[1] if len(sys.argv) > 1 else None
    tiles = [int(t) for t in os.environ.get('CONV_BENCH_TILES', '-1,8,11,12,16').split(',')]
    for name, cin, cout, k, s, p, tr, H in LAYERS:
        if only and only not in name:
            continue
        spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
        d = spec.desc(B, H, H)
        OH, OW = spec.out_hw(H, H)
        x = torch.randn(B, H, H, ops.c4(cin), device=dev)
        dy = torch.randn(B, OH, OW, ops.c4(cout), device=dev)
        x = x.bfloat16() if d.x_bf16 else x          # RGB tensors stay fp32 (NHWC4)
        dy = dy.bfloat16() if d.y_bf16 else dy
        wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
        w = torch.randn(*wshape, device=dev) * 0.05
        wf, wb = ops.conv_prep(spec, d, w, None, True, True)
        flops = 2.0 * B * (H * H if tr else OH * OW) * cin * cout * k * k
        row = dict(layer=name, gflop=round(flops / 1e9, 1))
        for t in tiles:
            _lib.call('iprgan_debug_force_tiles', t, -1)
            t_f = timeit(lambda: ops.conv_fwd(spec, d, x, wf, None, stats=cin > 4))
            # as inside a training step: fused activation derivative of the producer (reads the layer input) + column sums
            t_d = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb, x, 2, 0.1, colsums=True)) if cin > 4 else \
                timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb))
            row[f'fwd[{t}]'] = f'{t_f * 1e3:.0f}us {flops / t_f / 1e9:.0f}TF'
            row[f'dgrad[{t}]'] = f'{t_d * 1e3:.0f}us {flops / t_d / 1e9:.0f}TF'
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        for c in [int(t) for t in os.environ.get('CONV_BENCH_WGRAD', '-1,0,60,61,63,64,66,67').split(',')]:
            _lib.call('iprgan_debug_force_tiles', -1, c)
            t_w = timeit(lambda: ops.conv_bwd_weight(spec, d, x, dy, wshape, False))
            row[f'wgrad[{c}]'] = f'{t_w * 1e3:.0f}us {flops / t_w / 1e9:.0f}TF'
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
