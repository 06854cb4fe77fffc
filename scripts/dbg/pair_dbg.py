import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import numpy as np, torch
from oracle import nets, recipe
from iprgan import networks
dev = torch.device('cuda:0')
a, b, c = nets.SNDiscriminator64(), networks.SNDiscriminator64(), networks.SNDiscriminator64()
for n in (a, b, c): recipe.fill(n, 33); n.train()
b.to(dev); c.to(dev)
xr = torch.tanh(recipe.tensor(33, 1, (6, 3, 64, 64))); xf = torch.tanh(recipe.tensor(33, 2, (6, 3, 64, 64)))
ra, fa = a(xr), a(xf)
rb, fb = b.forward_pair(xr.to(dev), xf.to(dev))
rc, fc = c(xr.to(dev)), c(xf.to(dev))
for (k, va), (_, vb), (_, vc) in zip(a.state_dict().items(), b.state_dict().items(), c.state_dict().items()):
    if k.endswith(('weight_u', 'weight_v')):
        print(k, 'pair-vs-oracle', float((vb.cpu() - va).abs().max()), 'seq-vs-oracle', float((vc.cpu() - va).abs().max()), 'max', float(va.abs().max()))
print('logits', float((rb.cpu()-ra).abs().max()), float((fb.cpu()-fa).abs().max()), float((rc.cpu()-ra).abs().max()), float((fc.cpu()-fa).abs().max()))
