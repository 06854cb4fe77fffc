import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch, torch.nn.functional as F
from iprgan import ops, _lib
dev = torch.device('cuda:0')
torch.manual_seed(0)
B, cin, cout, H, W = 1, 3, 64, 16, 16
x = torch.randn(B, cin, H, W).bfloat16().float()
w = (torch.randn(cout, cin, 3, 3) * 0.1).bfloat16().float()
yr = F.conv2d(x, w, None, padding=1)
_lib.set_math('bf16')
spec = ops.ConvSpec(cin, cout, 3, 1, 1)
d = spec.desc(B, H, W)
wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
xn = torch.zeros(B, H, W, 4); xn[..., :3] = x.permute(0, 2, 3, 1)
y = ops.conv_fwd(spec, d, xn.to(dev), wf, None).cpu()          # [B,H,W,64]
ref = yr.permute(0, 2, 3, 1)
err = (y - ref).abs()
print('max err', float(err.max()), 'ref max', float(ref.abs().max()))
bad = (err > 1e-3)
print('bad frac', float(bad.float().mean()))
print('bad per channel', bad.float().mean((0, 1, 2)).tolist())
print('bad per row y', bad.float().mean((0, 2, 3)).tolist())
print('bad per col x', bad.float().mean((0, 1, 3)).tolist())
# try to find where value y[0,0,0,c] came from
for c in range(0, 16):
    v = float(y[0, 3, 5, c]); m = (ref[0] - v).abs() < 1e-4
    idx = m.nonzero()[:3].tolist()
    print(c, v, float(ref[0, 3, 5, c]), idx)
