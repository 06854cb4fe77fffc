"""Probe: which half of the DCGAN step replays wrongly?  usage: graph_probe2.py d|g|dg  (GPU only)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from iprgan import Config, graphs, models  # noqa: E402

dev = torch.device('cuda:0')
what = sys.argv[1] if len(sys.argv) > 1 else 'd'
B = 16
torch.manual_seed(3)
xs = [torch.tanh(torch.randn(B, 3, 64, 64, device=dev)) for _ in range(6)]
zs = [torch.randn(B, 128, device=dev) for _ in range(6)]


def build():
    torch.manual_seed(7)
    m = models.DCGAN(Config(bench.DCGAN_CFG), device=[dev])
    for o in (m.optG, m.optD):
        o.device_step = True
    return m


def body_of(m):
    def body(s):
        if 'd' in what:
            m.update_d({'real_sample': s['x'], 'latent': s['z']})
        else:
            m.forward_d({'real_sample': s['x'], 'latent': s['z']})
            m.compute_d_loss()
        if 'g' in what:
            m.update_g({'fake_sample': m.fake_sample})
    return body


def rec(m):
    return ['%.5f' % float(t.double().sum()) for t in (m.real_logits, m.fake_logits)] + \
           (['%.5f' % float(m.gen_logits.double().sum())] if 'g' in what else [])


a, b = build(), build()
ba = body_of(a)
ha, hb = [], []
for s in range(6):
    ba({'x': xs[s], 'z': zs[s]})
    ha.append(rec(a))
step = graphs.GraphedStep(b, body_of(b), {'x': xs[0], 'z': zs[0]}, warmup=2)
for s in range(6):
    step({'x': xs[s], 'z': zs[s]})
    hb.append(rec(b))
print(what, 'failed:', step.failed, 'replays', step.replays)
for s in range(6):
    print(s, ha[s], hb[s], 'OK' if ha[s] == hb[s] else 'DIFF')
