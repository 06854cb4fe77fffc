"""RGB stem forward (3->64 k3) at several sizes, fp32 and bf16 output storage: time and effective write bandwidth."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch
from iprgan import ops, _lib
dev = torch.device('cuda:0')


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for mode in ('fp32', 'bf16', 'bf16act'):
    _lib.set_math(mode)
    for B, H in ((128, 64), (256, 64), (128, 128), (256, 128)):
        spec = ops.ConvSpec(3, 64, 3, 1, 1, act=2, slope=0.1)
        d = spec.desc(B, H, H)
        x = torch.randn(B, H, H, 4, device=dev)
        w = torch.randn(64, 3, 3, 3, device=dev) * 0.1
        bias = torch.randn(64, device=dev)
        wf, _ = ops.conv_prep(spec, d, w, None, True, False)
        t = timeit(lambda: ops.conv_fwd(spec, d, x, wf, bias))
        ob = B * H * H * 64 * (2 if d.y_bf16 else 4)
        print(f'{mode:8s} B{B} {H}x{H}: {t:7.1f} us  out {ob / 1e6:7.1f} MB  {ob / t / 1e6:6.2f} TB/s write', flush=True)
_lib.set_math('fp32')
# reference: pure streaming writes / copies of the same size on this GPU
for mb in (134, 537, 1074):
    n = mb * 1000 * 1000 // 4
    buf = torch.empty(n, device=dev)
    src = torch.empty(n, device=dev)
    t = timeit(lambda: buf.fill_(1.0))
    t2 = timeit(lambda: buf.copy_(src))
    t3 = timeit(lambda: ops.fill(buf, 2.0)) if hasattr(ops, 'fill') else 0
    print(f'fill {mb} MB: torch {t:7.1f} us = {n * 4 / t / 1e6:5.2f} TB/s | iprgan_fill {t3:7.1f} us | copy {t2:7.1f} us = {2 * n * 4 / t2 / 1e6:5.2f} TB/s (r+w)')
