"""debug: full-tensor comparison of the step-0 Adam moments of the CycleGAN pool case (engine vs live oracle)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import numpy as np, torch
from oracle import cases, gan, recipe
from iprgan import Config, models

def run(make_cfg, mods, device):
    seed, batch, size = 53, 4, 64
    cfg = dict(cases.CYCLEGAN_CFG, pool_size=6, epoch=4)
    model = mods.CycleGAN(make_cfg(cfg), device=device)
    for i, n in enumerate((model.GA, model.GB, model.DA, model.DB)):
        recipe.fill(n.module, seed + i); n.to(device[0])
    a = torch.tanh(recipe.tensor(seed, 200, (batch, 3, size, size)))
    b = torch.tanh(recipe.tensor(seed, 300, (batch, 3, size, size)))
    model.update_g({'real_A': a, 'real_B': b})
    torch.manual_seed(900)
    model.update_d({'real_A': model.real_A, 'real_B': model.real_B, 'fake_A': model.fake_A.detach(), 'fake_B': model.fake_B.detach()})
    sd = model.state_dict()
    return {f'{o}/{i}': sd[o]['state'][i]['exp_avg'].detach().cpu().double() for o in ('optG', 'optD') for i in sorted(sd[o]['state'])}, model

ref, mo = run(gan.Cfg, gan, gan.CPU)
res, me = run(Config, models, [torch.device('cuda:0')])
names = [k for k, _ in list(mo.DA.named_parameters())] + [k for k, _ in list(mo.DB.named_parameters())]
for k in ref:
    a, b = res[k], ref[k]
    err = (a - b).abs()
    sc = float(b.abs().max())
    rel = err / (b.abs() + 1e-3 * sc)
    bad = int((rel > 5e-3).sum())
    print(f'{k:10s} shape {tuple(b.shape)} max|b| {sc:.3e} maxerr/max {float(err.max())/max(sc,1e-30):.2e} L2rel {float((a-b).norm()/b.norm()):.2e} bad {bad}/{b.numel()}')
    if bad and k.startswith('optD/0'):
        idx = torch.nonzero(rel > 5e-3)
        print('  bad idx (first 20):', idx[:20].tolist())
