"""Probe (GPU): what does the fused activation derivative (operand read in the backward-data epilogue) cost per layer of
BASELINE config 5?  Times backward-data with and without it, autotuned tiles."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
sys.path.insert(0, os.path.join(ROOT, 'scripts'))
import torch  # noqa: E402
from iprgan import _lib, ops  # noqa: E402
from conv_bench_bf16 import LAYERS, timeit, B  # noqa: E402

dev = torch.device('cuda:0')
_lib.set_math('bf16act')
for name, cin, cout, k, s, p, tr, H in LAYERS:
    if cin <= 4 or 'GEMM' in name:
        continue
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
    d = spec.desc(B, H, H)
    OH, OW = spec.out_hw(H, H)
    x = torch.randn(B, H, H, ops.c4(cin), device=dev).bfloat16()
    dy = torch.randn(B, OH, OW, ops.c4(cout), device=dev)
    dy = dy.bfloat16() if d.y_bf16 else dy
    w = torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), device=dev) * 0.05
    _, wb = ops.conv_prep(spec, d, w, None, False, True)
    t_aux = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb, x, 2, 0.1, colsums=True))
    t_cs = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb, colsums=True))
    t_plain = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb))
    print(f'{name:28s} dgrad + derivative + colsums {t_aux * 1e3:6.0f} us | colsums only {t_cs * 1e3:6.0f} us | plain {t_plain * 1e3:6.0f} us', flush=True)
