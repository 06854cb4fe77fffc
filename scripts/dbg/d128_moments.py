import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import numpy as np, torch
from oracle import cases, gan, recipe
from iprgan import Config, models

def run(make_cfg, mods, device):
    seed, batch, size = 91, 8, 128
    model = mods.DCGAN(make_cfg(cases.DCGAN128_CFG), device=device)
    recipe.fill(model.G.module, seed); recipe.fill(model.D.module, seed + 1)
    model.G.to(device[0]); model.D.to(device[0])
    model = mods.WhiteBoxWrapper(model, make_cfg(cases.WBOX_CFG))
    x = torch.tanh(recipe.tensor(seed, 2000, (batch, 3, size, size))); z = recipe.tensor(seed, 3000, (batch, 128))
    model.update_d({'real_sample': x, 'latent': z})
    model.update_g({'fake_sample': model.fake_sample})
    sd = model.state_dict()
    return {f'{o}/{i}': sd[o]['state'][i]['exp_avg'].detach().cpu().double() for o in ('optG', 'optD') for i in sorted(sd[o]['state'])}

ref = run(gan.Cfg, gan, gan.CPU)
res = run(Config, models, [torch.device('cuda:0')])
for k in ref:
    a, b = res[k], ref[k]
    print(f'{k:9s} {str(tuple(b.shape)):22s} max|b| {float(b.abs().max()):.2e} L2rel {float((a-b).norm()/b.norm()):.2e} maxerr/max {float((a-b).abs().max()/b.abs().max()):.2e}')
