"""BatchNorm kernels per tensor size and storage kind (run under rocprofv3 --kernel-trace --stats, one size per process):
python scripts/probe/bn_probe.py <B> <H> <C> <kind 0|2>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

B, H, C, kind = [int(v) for v in sys.argv[1:5]]
dev = torch.device('cuda:0')
_lib.set_math('fp32x3' if kind == 2 else 'fp32')
x = ops.to_kind(torch.randn(B, H, H, C, device=dev), kind)
dy = ops.to_kind(torch.randn(B, H, H, C, device=dev), kind)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
for _ in range(20):
    y, mean, invstd = ops.bn_fwd(x, g, b, rm, rv, 1e-5, 0.1, True, _lib.ACT_RELU)
    ops.bn_bwd(x, y, dy, g, mean, invstd, _lib.ACT_RELU, beta=b)
torch.cuda.synchronize()
print('done', B, H, C, kind, x.numel())
