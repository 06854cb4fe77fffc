"""Golden-case runners shared by the fixture generator and the parity tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  A runner takes an implementation
(the real reference, the oracle, or the HIP product) as a namespace of builders,
drives it on recipe-generated weights/inputs and returns {name: ndarray}.
``gen_golden.py`` runs them on the real reference and stores the result;
the tests run them on the oracle (CPU) and on the HIP engine (GPU) and compare.
"""
import numpy as np
import torch

from . import recipe

# name -> (builder attr, builder kwargs, input shape, seed)
NET_CASES = {
    'ConvGenerator64':   ('ConvGenerator64', {}, (2, 128), 11),
    'ConvGenerator32':   ('ConvGenerator32', {}, (3, 128), 12),
    'SNDiscriminator64': ('SNDiscriminator64', {}, (2, 3, 64, 64), 13),
    'SNDiscriminator32': ('SNDiscriminator32', {}, (3, 3, 32, 32), 14),
    'SRResNet':          ('SRResNet', {}, (2, 3, 12, 12), 15),
    'Discriminator96':   ('Discriminator96', {}, (2, 3, 96, 96), 16),
    'Resnet6Blocks':     ('Resnet6Blocks', {}, (1, 3, 32, 32), 17),
    'ConvDiscriminator': ('ConvDiscriminator', {}, (1, 3, 64, 64), 18),
    'Decoder32':         ('Decoder32', {}, (3, 128), 19),
    'Encoder32':         ('Encoder32', {}, (3, 3, 32, 32), 20),      # returns (z, (mean, logvar)); eps replayed
}


def replay_eps(net):
    """The reference draws the VAE's eps with torch.randn_like on the CPU generator (networks/encoder.py:26);
    an implementation exposing ``eps_fn`` (the HIP Encoder32) is pointed at the same CPU stream so that
    ``torch.manual_seed`` replays the identical draw on every side."""
    if 'eps_fn' in getattr(net, '__dict__', {}):
        net.__dict__['eps_fn'] = lambda shape, device: torch.randn(shape)


BUFFER_LEAVES = ('running_mean', 'running_var', 'num_batches_tracked', 'weight_u', 'weight_v')


def run_net_case(networks, name, device='cpu'):
    """Seeded weights -> train-mode forward -> loss=(out*R).sum() -> backward."""
    attr, kw, xshape, seed = NET_CASES[name]
    net = getattr(networks, attr)(**kw)
    recipe.fill(net, seed)
    net = net.to(device)
    net.train()
    x = recipe.tensor(seed, 1000, xshape).to(device)
    if len(xshape) == 4:
        x = torch.tanh(x)
    x.requires_grad_(True)
    replay_eps(net)
    torch.manual_seed(seed)
    out = net(x)
    if isinstance(out, tuple):                  # Encoder32
        out = torch.cat([out[0], out[1][0], out[1][1]], dim=1)
    r = recipe.tensor(seed, 1001, tuple(out.shape)).to(device)
    loss = (out * r).sum()
    loss.backward()
    res = {'out': out.detach().cpu().numpy(), 'loss': np.float64(loss.item())}
    recipe.pack_summary('dx', x.grad.cpu(), res)
    for k, p in net.named_parameters():
        recipe.pack_summary(f'grad/{k}', p.grad.cpu(), res)
    for k, b in net.state_dict().items():
        if k.rsplit('.', 1)[-1] in BUFFER_LEAVES:
            res[f'buf/{k}'] = b.detach().cpu().numpy()
    return res


DCGAN_CFG = {'G': 'ConvGenerator64', 'D': 'SNDiscriminator64', 'opt': 'Adam',
             'opt_param': {'lr': 2.0e-4, 'betas': [0.5, 0.999]}, 'type': 'DCGAN'}
WBOX_CFG = {'gamma_0': 0.1, 'string': 'EXAMPLE A', 'target': 'G'}


# BASELINE config 5: the 128x128 DCGAN.  The reference has no factory for it ("construct
# ConvGenerator(mg=16) / SNDiscriminator(md=16)"); every side registers these two names over its own classes.
DCGAN128_CFG = dict(DCGAN_CFG, G='ConvGenerator128', D='SNDiscriminator128')


def run_dcgan_steps(make_cfg, models, device, n_steps=3, batch=4, seed=21, wbox=True, cfg=None, size=64):
    """n G+D steps in the order of experiments/image_generation.py:86-101 (update_d on
    {real_sample, latent}, then update_g on {fake_sample: model.fake_sample}) with the
    white-box wrapper (configure_protection, image_generation.py:75-84)."""
    model = models.DCGAN(make_cfg(cfg or DCGAN_CFG), device=device)
    recipe.fill(model.G.module, seed)
    recipe.fill(model.D.module, seed + 1)
    model.G.to(device[0])
    model.D.to(device[0])
    if wbox:
        model = models.WhiteBoxWrapper(model, make_cfg(WBOX_CFG))
    res = {}
    for s in range(n_steps):
        x = torch.tanh(recipe.tensor(seed, 2000 + s, (batch, 3, size, size)))
        z = recipe.tensor(seed, 3000 + s, (batch, 128))
        model.update_d({'real_sample': x, 'latent': z})
        model.update_g({'fake_sample': model.fake_sample})
        for k, v in model.get_metrics().items():
            res[f'step{s}/metric/{k}'] = np.float64(v)
        if s == 0:
            # after ONE step the optimizer moments are (1-beta)*grad: a tight check of every gradient
            res['step0/fake_sample'] = model.fake_sample.detach().cpu().numpy()
            sd0 = model.state_dict()
            for opt in ('optG', 'optD'):
                for idx in sorted(sd0[opt]['state']):
                    recipe.pack_summary(f'step0/{opt}/{idx}/exp_avg', sd0[opt]['state'][idx]['exp_avg'].cpu(), res)
            for net in ('G', 'D'):
                for k, v in sd0[net].items():
                    if k.rsplit('.', 1)[-1] in BUFFER_LEAVES:
                        res[f'step0/{net}/{k}'] = v.detach().cpu().numpy().copy()
    sd = model.state_dict()
    for net in ('G', 'D'):
        for k, v in sd[net].items():
            leaf = k.rsplit('.', 1)[-1]
            v = v.detach().cpu()
            if leaf in BUFFER_LEAVES or v.numel() <= 512:
                res[f'final/{net}/{k}'] = v.numpy()
            else:
                recipe.pack_summary(f'final/{net}/{k}', v, res)
    for opt in ('optG', 'optD'):
        st = sd[opt]['state']
        for idx in sorted(st):
            recipe.pack_summary(f'final/{opt}/{idx}/exp_avg', st[idx]['exp_avg'].cpu(), res)
            recipe.pack_summary(f'final/{opt}/{idx}/exp_avg_sq', st[idx]['exp_avg_sq'].cpu(), res)
            res[f'final/{opt}/{idx}/step'] = np.float64(float(st[idx]['step']))
    for k, v in sd['sign'].items() if wbox else ():
        res[f'final/sign/{k}'] = v.detach().cpu().numpy()
    if wbox:
        res['final/ber'] = np.float64(float(model.loss_model.compute_ber(model.G)))
    return res


SIGN_CASES = {'ConvGenerator64': 448, 'SRResNet': 2112, 'Resnet9Blocks': 5248}


def run_sign_case(networks, SignLossModel, make_cfg, name, string='EXAMPLE A', seed=31):
    """Bits/signs assigned per norm layer, the sign-loss value and the BER after a
    deterministic corruption of gamma (zeros count as errors, sign_model.py:58)."""
    net = getattr(networks, name)()
    recipe.fill(net, seed)
    slm = SignLossModel(net, make_cfg({'gamma_0': 0.1, 'string': string}))
    slm = slm.to(next(net.parameters()).device)      # as models/wrappers.py:83-86 does
    res = {}
    signs = [b.detach().cpu().numpy() for _, b in slm.named_buffers()]
    res['signs'] = np.concatenate(signs).astype(np.int8)
    res['names'] = np.array([k for k, _ in slm.named_buffers()])
    res['ber_clean'] = np.float64(float(slm.compute_ber(net)))
    res['loss_clean'] = np.float64(float(slm(net).detach()))
    g = np.random.default_rng([seed, 77])
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.InstanceNorm2d)) and m.weight is not None:
                n = m.weight.numel()
                flip = torch.from_numpy(g.random(n) < 0.2)
                zero = torch.from_numpy(g.random(n) < 0.05)
                m.weight[flip] = -m.weight[flip]
                m.weight[zero] = 0.0
    res['ber_corrupt'] = np.float64(float(slm.compute_ber(net)))
    res['loss_corrupt'] = np.float64(float(slm(net).detach()))
    return res


def _capture(model, res, tag, nets, opts, moments=True):
    sd = model.state_dict()
    for net in nets:
        for k, v in sd[net].items():
            leaf = k.rsplit('.', 1)[-1]
            v = v.detach().cpu()
            if leaf in BUFFER_LEAVES or v.numel() <= 512:
                res[f'{tag}/{net}/{k}'] = v.numpy().copy()
            else:
                recipe.pack_summary(f'{tag}/{net}/{k}', v, res)
    if moments:
        for opt in opts:
            st = sd[opt]['state']
            for idx in sorted(st):
                recipe.pack_summary(f'{tag}/{opt}/{idx}/exp_avg', st[idx]['exp_avg'].cpu(), res)


BBOX_CFG = {'fn_inp': {'type': 'TransformDist'}, 'fn_out': {'type': 'RandomNoisePatch', 'size': 16},
            'lambda': 1.0, 'loss_fn': 'ssim',
            # set by ImageGeneration.configure_protection (experiments/image_generation.py:63-67)
            'normalized': True, 'input_var': 'latent', 'output_var': 'generated', 'target': 'G'}


def run_bbox_transforms(tools, make_cfg):
    """Known answers of the trigger transforms (tools/transform_dist.py, random_bitmask.py, transform_var.py):
    seeded construction, fixed latent batch."""
    z = recipe.tensor(71, 1, (5, 128))
    res = {'dist': tools.TransformDist(make_cfg({}))(z).numpy()}
    torch.manual_seed(72)
    m = tools.RandomBitMask(make_cfg({'n_bit': 10, 'constant': -10.0, 'z_dim': 128}))
    res['bitmask/mask'] = m._mask.numpy().astype(np.int64)
    res['bitmask/out'] = m(z).numpy()
    torch.manual_seed(73)
    v = tools.TransformVar(make_cfg({}))
    res['var/w'], res['var/a'], res['var/out'] = v.w.numpy(), v.a.numpy(), v(z).numpy()
    return res


def run_loss_factories(loss_ns):
    """tools.l1 / tools.mse (tools/loss.py:10-20,72-76) with and without ``normalized``: value and gradient w.r.t. x on
    a seeded image pair in [-1, 1] with a block of exact ties (sign(0) = 0 in the L1 gradient).  ``loss_ns``: anything
    with ``l1`` / ``mse`` factories (the real tools/loss.py, oracle.bbox, iprgan.tools)."""
    x = torch.tanh(recipe.tensor(5, 1, (3, 3, 20, 24)))
    y = torch.tanh(recipe.tensor(5, 2, (3, 3, 20, 24)))
    y[0, 0, :2] = x[0, 0, :2]
    res = {}
    for name in ('l1', 'mse'):
        for normalized in (False, True):
            dev = getattr(loss_ns, 'DEVICE', None)
            xa = (x.to(dev) if dev is not None else x.clone()).requires_grad_()
            loss = getattr(loss_ns, name)(normalized=normalized)(xa, y.to(dev) if dev is not None else y)
            loss.backward()
            key = f'{name}/{"norm" if normalized else "raw"}'
            res[key + '/value'] = np.float64(float(loss.detach()))
            res[key + '/grad'] = xa.grad.detach().cpu().numpy()
    return res


def run_dcgan_complete_steps(make_cfg, models, device, n_steps=2, batch=4, seed=81):
    """The 'complete' protection (configs/DCGAN/complete/*.yaml): BlackBoxWrapper (TransformDist trigger ->
    noise-patch target, SSIM) inside WhiteBoxWrapper, in the order of experiments/image_generation.py:56-101."""
    model = models.DCGAN(make_cfg(DCGAN_CFG), device=device)
    recipe.fill(model.G.module, seed)
    recipe.fill(model.D.module, seed + 1)
    model.G.to(device[0])
    model.D.to(device[0])
    torch.manual_seed(seed)                         # RandomNoisePatch draws its patch at construction
    model = models.BlackBoxWrapper(model, make_cfg(BBOX_CFG))
    model = models.WhiteBoxWrapper(model, make_cfg(WBOX_CFG))
    res = {}
    for s in range(n_steps):
        x = torch.tanh(recipe.tensor(seed, 2000 + s, (batch, 3, 64, 64)))
        z = recipe.tensor(seed, 3000 + s, (batch, 128))
        model.update_d({'real_sample': x, 'latent': z})
        model.update_g({'fake_sample': model.fake_sample})
        for k, v in model.get_metrics().items():
            res[f'step{s}/metric/{k}'] = np.float64(v)
        if s == 0:
            res['step0/fake_sample'] = model.fake_sample.detach().cpu().numpy()
            res['step0/xwm'] = model.xwm.detach().cpu().numpy()
            res['step0/ywm'] = model.ywm.detach().cpu().numpy()
            res['step0/Gxwm'] = model.Gxwm.detach().cpu().numpy()
            _capture(model, res, 'step0', ('G',), ('optG', 'optD'))
    _capture(model, res, 'final', ('G', 'D', 'fn_out'), ('optG', 'optD'))
    res['final/ber'] = np.float64(float(model.loss_model.compute_ber(model.G)))
    return res


VAE_CFG = {'G': 'Decoder32', 'D': 'Encoder32', 'opt': 'Adam',
           'opt_param': {'lr': 3.0e-5, 'weight_decay': 1.0e-6}, 'type': 'VAE'}      # configs/VAE/*/vae-cifar10-a.yaml


def run_vae_steps(make_cfg, models, device, n_steps=3, batch=4, seed=61, wbox=True):
    """VAE steps in the order of experiments/image_generation.py:86-101: update_d (forward only, models/vae.py:66-67)
    then update_g (loss, backward, ONE Adam over decoder + encoder), white-box sign loss on the decoder's BN."""
    model = models.VAE(make_cfg(VAE_CFG), device=device)
    recipe.fill(model.G.module, seed)
    recipe.fill(model.D.module, seed + 1)
    model.G.to(device[0])
    model.D.to(device[0])
    replay_eps(model.D.module)
    if wbox:
        model = models.WhiteBoxWrapper(model, make_cfg(WBOX_CFG))
    res = {}
    for s in range(n_steps):
        x = torch.tanh(recipe.tensor(seed, 2000 + s, (batch, 3, 32, 32)))
        torch.manual_seed(7000 + s)
        model.update_d({'real_sample': x, 'latent': None})
        model.update_g({'fake_sample': model.fake_sample})
        for k, v in model.get_metrics().items():
            res[f'step{s}/metric/{k}'] = np.float64(float(v.detach()) if torch.is_tensor(v) else float(v))
        if s == 0:
            res['step0/fake_sample'] = model.fake_sample.detach().cpu().numpy()
            res['step0/latent'] = model.latent.detach().cpu().numpy()
            _capture(model, res, 'step0', (), ('opt',))
    _capture(model, res, 'final', ('G', 'D'), ('opt',))
    if wbox:
        for k, v in model.state_dict()['sign'].items():
            res[f'final/sign/{k}'] = v.detach().cpu().numpy()
        res['final/ber'] = np.float64(float(model.loss_model.compute_ber(model.G)))
    return res


SRGAN_CFG = {'G': 'SRResNet', 'D': 'Discriminator96', 'V': 'VGG19Feature', 'opt': 'Adam',
             'opt_param': {'lr': 1.0e-4}, 'type': 'SRGAN'}


def bbox_cfg(kind, watermark, inp_size, out_size):
    """protection.bbox of configs/{SRGAN,CycleGAN}/complete/*.yaml + the keys the experiments add
    (image_super_resolution.py:62-65, image_translation.py:68-71)."""
    extra = {'super_resolution': (False, 'low_res', 'super_res', 'G'),
             'translation': (True, 'real_B', 'fake_A', 'GB')}[kind]
    return {'fn_inp': {'type': 'RandomNoisePatch', 'size': inp_size},
            'fn_out': {'type': 'PasteWatermark', 'size': out_size, 'opaque': True, 'watermark': watermark},
            'lambda': 1.0, 'loss_fn': 'ssim',
            'normalized': extra[0], 'input_var': extra[1], 'output_var': extra[2], 'target': extra[3]}


def run_srgan_steps(make_cfg, models, device, batch=2, seed=41, bbox=None):
    """One pre-training step (pixel MSE) then one GAN-phase G step and D step, in the order of
    experiments/image_super_resolution.py:84-113, with the white-box wrapper on G (and, when ``bbox`` is
    given, the black-box wrapper inside it as in configs/SRGAN/complete)."""
    model = models.SRGAN(make_cfg(SRGAN_CFG), device=device)
    recipe.fill(model.G.module, seed)
    recipe.fill(model.D.module, seed + 1)
    recipe.fill(model.V.module, seed + 2)
    for n in (model.G, model.D, model.V):
        n.to(device[0])
    if bbox:
        torch.manual_seed(seed)
        model = models.BlackBoxWrapper(model, make_cfg(bbox))
    model = models.WhiteBoxWrapper(model, make_cfg(WBOX_CFG))
    res = {}
    lr = recipe.tensor(seed, 100, (batch, 3, 24, 24), dist='uniform')
    hr = recipe.tensor(seed, 101, (batch, 3, 96, 96), dist='uniform')
    model.update_g({'low_res': lr, 'high_res': hr, 'pretrain': True, 'inhibit_bbox': True})
    for k, v in model.get_metrics().items():
        res[f'step0/metric/{k}'] = np.float64(v)
    res['step0/super_res'] = model.super_res.detach().cpu().numpy()
    _capture(model, res, 'step0', ('G',), ('optG',))
    lr = recipe.tensor(seed, 102, (batch, 3, 24, 24), dist='uniform')
    hr = recipe.tensor(seed, 103, (batch, 3, 96, 96), dist='uniform')
    model.update_g({'low_res': lr, 'high_res': hr, 'pretrain': False})
    model.update_d({'high_res': model.high_res, 'super_res': model.super_res})
    for k, v in model.get_metrics().items():
        res[f'step1/metric/{k}'] = np.float64(v)
    _capture(model, res, 'final', ('G', 'D'), ('optG', 'optD'))
    res['final/ber'] = np.float64(float(model.loss_model.compute_ber(model.G)))
    return res


CYCLEGAN_CFG = {'G': 'Resnet6Blocks', 'D': 'ConvDiscriminator', 'opt': 'Adam',
                'opt_param': {'lr': 2.0e-4, 'betas': [0.5, 0.999]}, 'type': 'CycleGAN', 'pool_size': 50,
                'lambda_A': 10.0, 'lambda_B': 10.0, 'lambda_idt': 0.5, 'epoch': 200}


def run_cyclegan_steps(make_cfg, models, device, n_steps=2, batch=1, size=64, seed=51, bbox=None, G=None):
    """G step then D step per iteration (experiments/image_translation.py:90-112), white-box on GB
    (black-box inside it when ``bbox`` is given, configs/CycleGAN/complete)."""
    model = models.CycleGAN(make_cfg(dict(CYCLEGAN_CFG, G=G) if G else CYCLEGAN_CFG), device=device)
    for i, n in enumerate((model.GA, model.GB, model.DA, model.DB)):
        recipe.fill(n.module, seed + i)
        n.to(device[0])
    if bbox:
        torch.manual_seed(seed)
        model = models.BlackBoxWrapper(model, make_cfg(bbox))
    wcfg = dict(WBOX_CFG)
    wcfg['target'] = 'GB'                            # image_translation.py:83
    model = models.WhiteBoxWrapper(model, make_cfg(wcfg))
    res = {}
    for s in range(n_steps):
        a = torch.tanh(recipe.tensor(seed, 200 + s, (batch, 3, size, size)))
        b = torch.tanh(recipe.tensor(seed, 300 + s, (batch, 3, size, size)))
        model.update_g({'real_A': a, 'real_B': b})
        model.update_d({'real_A': model.real_A, 'real_B': model.real_B,
                        'fake_A': model.fake_A.detach(), 'fake_B': model.fake_B.detach()})
        for k, v in model.get_metrics().items():
            res[f'step{s}/metric/{k}'] = np.float64(v)
        if s == 0:
            res['step0/fake_B'] = model.fake_B.detach().cpu().numpy()
            _capture(model, res, 'step0', ('GA', 'GB', 'DA', 'DB'), ('optG', 'optD'))
    _capture(model, res, 'final', ('GA', 'GB', 'DA', 'DB'), (), moments=False)
    res['final/poolA/images'] = model.state_dict()['poolA']['images'].detach().cpu().numpy()
    res['final/ber'] = np.float64(float(model.loss_model.compute_ber(model.GB)))
    return res


def run_cyclegan_pool_steps(make_cfg, models, device, n_steps=4, batch=4, size=64, seed=53):
    """CycleGAN at batch 4 with a SMALL history pool and a SHORT schedule so that four steps reach what the
    two-step fixture never does: ImagePool's swap branch once ``counts >= pool_size`` (models/util.py:27-34; the
    swap decisions are torch.rand / torch.randperm draws on the CPU generator, seeded per step here) and the
    linear learning-rate decay of ``update_lr()`` (models/cyclegan.py:50-56,145-147: epoch 4 -> factor 1, 1, 1,
    0.5).  At batch 4 the step-0 Adam moments are no longer dominated by single-pixel sign flips and are
    compared tightly."""
    cfg = dict(CYCLEGAN_CFG, pool_size=6, epoch=4)
    model = models.CycleGAN(make_cfg(cfg), device=device)
    for i, n in enumerate((model.GA, model.GB, model.DA, model.DB)):
        recipe.fill(n.module, seed + i)
        n.to(device[0])
    wcfg = dict(WBOX_CFG)
    wcfg['target'] = 'GB'
    model = models.WhiteBoxWrapper(model, make_cfg(wcfg))
    res = {}
    for s in range(n_steps):
        a = torch.tanh(recipe.tensor(seed, 200 + s, (batch, 3, size, size)))
        b = torch.tanh(recipe.tensor(seed, 300 + s, (batch, 3, size, size)))
        model.update_g({'real_A': a, 'real_B': b})
        torch.manual_seed(900 + s)                   # ImagePool draws from the CPU generator
        model.update_d({'real_A': model.real_A, 'real_B': model.real_B,
                        'fake_A': model.fake_A.detach(), 'fake_B': model.fake_B.detach()})
        for k, v in model.get_metrics().items():
            res[f'step{s}/metric/{k}'] = np.float64(v)
        res[f'step{s}/pool_counts'] = np.float64(float(model.state_dict()['poolA']['counts']))
        if s == 0:
            res['step0/fake_B'] = model.fake_B.detach().cpu().numpy()
            _capture(model, res, 'step0', ('GA', 'GB', 'DA', 'DB'), ('optG', 'optD'))
        model.update_lr()                            # once per epoch in image_translation.py; per step here
    _capture(model, res, 'final', ('GA', 'GB', 'DA', 'DB'), (), moments=False)
    sd = model.state_dict()
    res['final/poolA/images'] = sd['poolA']['images'].detach().cpu().numpy()
    res['final/poolB/images'] = sd['poolB']['images'].detach().cpu().numpy()
    res['final/lr'] = np.float64(model.optG.param_groups[0]['lr'])
    res['final/ber'] = np.float64(float(model.loss_model.compute_ber(model.GB)))
    return res


def vgg_layer_names_from_source(path):
    """The ``layer_name`` list literal of the reference's networks/vgg.py:6-28, read with ``ast`` (the file cannot be
    imported: torchvision is absent).  Only the list of names - data - is returned and stored as a fixture."""
    import ast
    tree = ast.parse(open(path).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(getattr(t, 'id', None) == 'layer_name' for t in node.targets):
            return [ast.literal_eval(e) for e in node.value.elts]
    raise RuntimeError('layer_name not found')


def vgg_layer_names(net_cls):
    """Names of an implementation's VGG19 feature stack in the reference's scheme (conv{b}_{i} / relu{b}_{i} /
    pool{b}), derived from the MODULE TYPES of the full 'pool5' stack it builds."""
    names, blk, idx = [], 1, 1
    for m in net_cls(layer='pool5').net:
        if isinstance(m, torch.nn.Conv2d):
            assert m.kernel_size == (3, 3) and m.padding == (1, 1) and m.stride == (1, 1)
            names.append(f'conv{blk}_{idx}')
        elif isinstance(m, torch.nn.ReLU):
            names.append(f'relu{blk}_{idx}')
            idx += 1
        elif isinstance(m, torch.nn.MaxPool2d):
            names.append(f'pool{blk}')
            blk, idx = blk + 1, 1
        else:
            raise AssertionError(f'unexpected module {m}')
    return names


# ---- late steps from a COMMON state (round 5) -----------------------------------------------------------------------
# After two or three optimizer steps at batch 1-4 two correct fp32 implementations have drifted apart chaotically: measured
# in round 5 (scripts/probe/final_moment_dist.py), the moments after the last step of the SRGAN / VAE fixtures sit up to 2-3x
# a tensor's scale away from the reference's IN BOTH math modes, the exact fp32 MFMA included - the golden comparison of
# `final/opt*` can only be one of overall magnitude.  What CAN be checked tightly at a late step is the step itself: both
# implementations start it from the SAME state (the leader's weights, buffers, Adam moments and step counts, loaded into the
# follower through the reference's own state_dict layout), run it on the same inputs, and every moment is compared
# element-wise.  A defect in a late-step moment (bias correction with step > 1, exp_avg_sq accumulation, a stale operand
# cache, a gradient written twice) shows up there at the step-0 tolerances.
def _cpu_state(sd):
    import copy
    def conv(v):
        if torch.is_tensor(v):
            return v.detach().cpu().clone()
        if isinstance(v, dict):
            return type(v)((k, conv(x)) for k, x in v.items())
        if isinstance(v, (list, tuple)):
            return type(v)(conv(x) for x in v)
        return copy.deepcopy(v)
    return conv(sd)


def _moments_full(model, opts):
    out, sd = {}, model.state_dict()
    for opt in opts:
        st = sd[opt]['state']
        for idx in sorted(st):
            out[f'{opt}/{idx}/exp_avg'] = st[idx]['exp_avg'].detach().cpu().double().numpy()
            out[f'{opt}/{idx}/exp_avg_sq'] = st[idx]['exp_avg_sq'].detach().cpu().double().numpy()
            out[f'{opt}/{idx}/step'] = np.float64(float(st[idx]['step']))
    return out


def run_late_step_pair(kind, lead, follow, lead_steps=2, truth64=False):
    """``lead`` / ``follow`` = (make_cfg, models namespace, device list).  The leader runs ``lead_steps`` steps of the ``kind``
    fixture ('dcgan', 'srgan', 'cyclegan'); the follower loads the leader's state; both run ONE more step on the same inputs.
    Returns (leader moments, follower moments, leader metrics, follower metrics) of that step, full tensors.
    truth64=True: a THIRD model - the follower's implementation with every network in float64 - loads the same state and runs
    the same step: its moments (the exact ones, up to 1e-15) are returned as a fifth value, so that the caller can hold the
    leader to a multiple of the follower's OWN distance from the truth instead of to bands fitted to observed runs."""
    def build(impl):
        make_cfg, models, device = impl
        if kind == 'dcgan':
            m = models.DCGAN(make_cfg(DCGAN_CFG), device=device)
            nets_, seed = (m.G, m.D), 21
            wcfg = dict(WBOX_CFG)
        elif kind == 'srgan':
            m = models.SRGAN(make_cfg(SRGAN_CFG), device=device)
            nets_, seed = (m.G, m.D, m.V), 41
            wcfg = dict(WBOX_CFG)
        else:
            m = models.CycleGAN(make_cfg(CYCLEGAN_CFG), device=device)
            nets_, seed = (m.GA, m.GB, m.DA, m.DB), 51
            wcfg = dict(WBOX_CFG, target='GB')
        for i, n in enumerate(nets_):
            recipe.fill(n.module, seed + i)
            n.to(device[0])
        return models.WhiteBoxWrapper(m, make_cfg(wcfg)), seed

    def step(model, seed, s, cast=lambda v: v):
        if kind == 'dcgan':
            x = cast(torch.tanh(recipe.tensor(seed, 2000 + s, (4, 3, 64, 64))))
            z = cast(recipe.tensor(seed, 3000 + s, (4, 128)))
            model.update_d({'real_sample': x, 'latent': z})
            model.update_g({'fake_sample': model.fake_sample})
        elif kind == 'srgan':
            lr = cast(recipe.tensor(seed, 100 + 2 * s, (2, 3, 24, 24), dist='uniform'))
            hr = cast(recipe.tensor(seed, 101 + 2 * s, (2, 3, 96, 96), dist='uniform'))
            if s == 0:
                model.update_g({'low_res': lr, 'high_res': hr, 'pretrain': True, 'inhibit_bbox': True})
            else:
                model.update_g({'low_res': lr, 'high_res': hr, 'pretrain': False})
                model.update_d({'high_res': model.high_res, 'super_res': model.super_res})
        else:
            a = cast(torch.tanh(recipe.tensor(seed, 200 + s, (1, 3, 64, 64))))
            b = cast(torch.tanh(recipe.tensor(seed, 300 + s, (1, 3, 64, 64))))
            model.update_g({'real_A': a, 'real_B': b})
            model.update_d({'real_A': model.real_A, 'real_B': model.real_B,
                            'fake_A': model.fake_A.detach(), 'fake_B': model.fake_B.detach()})
        return {k: float(v) for k, v in model.get_metrics().items()}

    a, seed = build(lead)
    b, _ = build(follow)
    for s in range(lead_steps):
        step(a, seed, s)
    state = _cpu_state(a.state_dict())
    b.load_state_dict(state, strict=True)
    t64 = None
    if truth64:
        t, _ = build(follow)
        _to_double(t)
        t.load_state_dict(_cpu_state(state), strict=True)      # (torch's optimizer loader casts the moments to the parameters' dtype)
        _to_double(t)
    ma, mb = step(a, seed, lead_steps), step(b, seed, lead_steps)
    if truth64:
        step(t, seed, lead_steps, cast=lambda v: v.double())
        t64 = _moments_full(t, ('optG', 'optD'))
        return _moments_full(a, ('optG', 'optD')), _moments_full(b, ('optG', 'optD')), ma, mb, t64
    return _moments_full(a, ('optG', 'optD')), _moments_full(b, ('optG', 'optD')), ma, mb


def _to_double(model):
    """Every network (and floating-point buffer) of an oracle model in float64, through the wrappers (model.model ...)."""
    seen = set()
    m = model
    while m is not None and id(m) not in seen:
        seen.add(id(m))
        for v in list(getattr(m, '_modules', {}).values()) + list(m.__dict__.values()):
            if isinstance(v, torch.nn.Module):
                v.double()
        m = m.__dict__.get('model', None) or getattr(m, '_modules', {}).get('model', None)
