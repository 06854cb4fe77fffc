"""Probe: does a captured D forward (spectral norm advancing in place) replay like eager calls?  GPU only."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from iprgan import networks  # noqa: E402

dev = torch.device('cuda:0')
torch.manual_seed(0)
which = sys.argv[1] if len(sys.argv) > 1 else 'D'


def build():
    torch.manual_seed(1)
    net = networks.SNDiscriminator64() if which == 'D' else networks.ConvGenerator64()
    return net.to(dev).train()


x = torch.tanh(torch.randn(16, 3, 64, 64, device=dev)) if which == 'D' else torch.randn(16, 128, device=dev)
a, b = build(), build()
with torch.no_grad():
    ref = [float(a(x).double().sum()) for _ in range(6)]
    outs = [float(b(x).double().sum()) for _ in range(2)]        # warm-up (eager)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = b(x)
    for _ in range(4):
        g.replay()
        outs.append(float(y.double().sum()))
print('eager ', ['%.6f' % v for v in ref])
print('graph ', ['%.6f' % v for v in outs])

# ---- the same with the weights changed in place between calls (what an optimizer step does)
a, b = build(), build()
with torch.no_grad():
    ref, outs = [], []
    for i in range(6):
        ref.append(float(a(x).double().sum()))
        for p in a.parameters():
            p.mul_(1.01)
    for i in range(2):
        outs.append(float(b(x).double().sum()))
        for p in b.parameters():
            p.mul_(1.01)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = b(x)
    for _ in range(4):
        g.replay()
        outs.append(float(y.double().sum()))
        for p in b.parameters():
            p.mul_(1.01)
print('eager+upd ', ['%.6f' % v for v in ref])
print('graph+upd ', ['%.6f' % v for v in outs])
