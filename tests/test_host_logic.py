"""CPU: host-side logic of the engine that needs no kernel - module trees / state_dict layout,
sign-bit assignment, config object, wrapper delegation, optimizer state layout, loud failure."""
import numpy as np
import pytest
import torch

from oracle import cases, gan, nets, recipe, sign


def test_product_networks_mirror_reference_state_dict():
    from iprgan import networks
    for name in ('ConvGenerator32', 'ConvGenerator64', 'SNDiscriminator32', 'SNDiscriminator64', 'Encoder32', 'Decoder32'):
        a, b = getattr(nets, name)(), getattr(networks, name)()
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa) == list(sb), name
        assert [tuple(v.shape) for v in sa.values()] == [tuple(v.shape) for v in sb.values()]
        assert [n for n, _ in a.named_modules()] == [n for n, _ in b.named_modules()]
        b.load_state_dict(sa)                                   # interchangeable checkpoints


def test_bitgenerator_matches_reference_bits(golden):
    from iprgan import tools
    bits = tools.BitGenerator('EXAMPLE A').get(200)
    assert np.array_equal(np.array(bits, dtype=np.int8), golden('bits_EXAMPLE_A')['bits'])
    assert tools.BitGenerator(None).get(5) is not None          # random mode (sign_model.py:16-17)


def test_create_signs_bit_exact_on_product_generator(golden):
    from iprgan import Config, networks, tools
    net = networks.ConvGenerator64()
    recipe.fill(net, 31)
    slm = tools.SignLossModel(net, Config({'gamma_0': 0.1, 'string': 'EXAMPLE A'}))
    ref = golden('sign_ConvGenerator64')
    signs = np.concatenate([b.numpy() for _, b in slm.named_buffers()]).astype(np.int8)
    assert np.array_equal(signs, ref['signs'])
    assert [k for k, _ in slm.named_buffers()] == list(ref['names'])
    for (_, b), m in zip(slm.named_buffers(), [m for m in net.modules() if isinstance(m, torch.nn.BatchNorm2d)]):
        assert torch.equal(torch.sign(m.weight.data), b)        # gamma <- |gamma| * bit


def test_config_semantics(tmp_path):
    from iprgan import Config
    c = Config({'a': 1, 'b': {'c': [1, 2], 'd': {'e': 'x'}}})
    assert c.a == 1 and c['b'].c == [1, 2] and c.b.d.e == 'x' and c.get('zz', 7) == 7
    c.b['c'] = 5
    c['n'] = 3
    assert c.to_dict() == {'a': 1, 'b': {'c': 5, 'd': {'e': 'x'}}, 'n': 3}
    p = tmp_path / 'c.yaml'
    p.write_text(c.to_yaml())
    assert Config.parse(str(p)).to_dict() == c.to_dict()


def test_engine_block_of_the_config(tmp_path, monkeypatch):
    """The optional ``engine:`` block (configs.apply_engine): applied to the environment and the import-time switches;
    typos and bad values are errors; a config without the block changes nothing."""
    import os
    from iprgan import Config, configs, engine, models
    for k in ('IPRGAN_BUCKET_MB', 'IPRGAN_COMM', 'IPRGAN_TUNE_CACHE', 'IPRGAN_FUSE_STATS', 'IPRGAN_PAIR_D'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(engine, '_FUSE_STATS', True)
    monkeypatch.setattr(models, '_PAIR_D', True)
    f = tmp_path / 'c.yaml'
    f.write_text('experiment: ImageGeneration\nengine:\n  math: bf16act\n  bucket_mb: 16\n  comm: torch\n'
                 '  tune_cache: /tmp/t.json\n  fuse_stats: false\n  pair_d: false\n')
    cfg = Config.parse(str(f))
    applied = configs.apply_engine(cfg, set_math=False)          # (the math mode itself is a library call: GPU tests)
    assert applied['math'] == 'bf16act' and applied['bucket_mb'] == 16
    assert os.environ['IPRGAN_BUCKET_MB'] == '16' and os.environ['IPRGAN_COMM'] == 'torch'
    assert os.environ['IPRGAN_TUNE_CACHE'] == '/tmp/t.json' and os.environ['IPRGAN_FUSE_STATS'] == '0'
    assert engine._FUSE_STATS is False and models._PAIR_D is False
    assert configs.apply_engine(Config({'experiment': 'x'})) == {}
    with pytest.raises(ValueError):
        configs.apply_engine(Config({'engine': {'mathh': 'fp32'}}))
    with pytest.raises(ValueError):
        configs.apply_engine(Config({'engine': {'math': 'fp8'}}), set_math=False)
    with pytest.raises(ValueError):
        configs.apply_engine(Config({'engine': {'comm': 'mpi'}}))


def test_models_build_on_cpu_but_refuse_to_compute():
    from iprgan import Config, models
    m = models.DCGAN(Config(cases.DCGAN_CFG))
    w = models.WhiteBoxWrapper(m, Config(cases.WBOX_CFG))
    assert list(w.state_dict()) == ['G', 'D', 'optG', 'optD', 'sign']
    assert w.G is m.G and w.no_such_attribute is None           # models/base.py:52-58
    assert all(k.startswith('module.') for k in m.G.state_dict())
    assert list(w.state_dict()['sign']) == ['module_convs_0_1', 'module_convs_1_1', 'module_convs_2_1']
    with pytest.raises(RuntimeError, match='no CPU'):
        w.update_d({'real_sample': torch.zeros(2, 3, 64, 64), 'latent': torch.zeros(2, 128)})


def test_adam_state_dict_layout_is_torch_compatible():
    from iprgan import optim
    p = [torch.nn.Parameter(torch.zeros(3, 3)), torch.nn.Parameter(torch.zeros(4))]
    mine = optim.Adam(p, lr=2e-4, betas=[0.5, 0.999])
    ref = torch.optim.Adam(p, lr=2e-4, betas=(0.5, 0.999))
    a, b = mine.state_dict()['param_groups'][0], ref.state_dict()['param_groups'][0]
    for k in ('lr', 'betas', 'eps', 'weight_decay', 'params'):
        assert a[k] == b[k], k
    for q in p:
        q.grad = torch.ones_like(q)
    ref.step()
    mine.load_state_dict(ref.state_dict())                      # a torch.optim.Adam checkpoint loads
    assert set(mine.state_dict()['state'][0]) >= {'step', 'exp_avg', 'exp_avg_sq'}


def test_removal_attacks_and_ber_counts():
    """sign_flip.py:59-75 / prune.py:46-57 restated as library functions: after flipping k of n scales the
    bit-error rate is exactly k/n; pruning zeroes scales (sign 0 counts as an error) below a global percentile."""
    from iprgan import attacks
    net = nets.ConvGenerator64()
    recipe.fill(net, 3)
    model = sign.SignLossModel(net, gan.Cfg({'gamma_0': 0.1, 'string': 'EXAMPLE A'}))
    assert float(model.compute_ber(net)) == 0.0
    g = torch.Generator().manual_seed(7)
    mask = attacks.sign_flip_(net, 30, generator=g)
    n = sum(w.numel() for w in attacks.norm_scales(net))
    assert n == 448 and int((mask < 0).sum()) == int(448 * 30 / 100)
    assert float(model.compute_ber(net)) == float(np.float32(134) / np.float32(448))      # fp32 ratio of exact counts
    attacks.sign_flip_(net, 100)                                # flip everything: previous flips come back
    assert float(model.compute_ber(net)) == float(np.float32(448 - 134) / np.float32(448))
    # prune: threshold is the percentile over ALL state_dict entries, like the script
    net2 = nets.ConvGenerator64()
    recipe.fill(net2, 3)
    model2 = sign.SignLossModel(net2, gan.Cfg({'gamma_0': 0.1, 'string': 'EXAMPLE A'}))
    sd = net2.state_dict()
    flat = np.concatenate([v.abs().double().numpy().ravel() for v in sd.values()])
    thr = attacks.prune_(sd, 40)
    assert thr == float(np.percentile(flat, 40))
    zeroed = sum(int((w == 0).sum()) for w in attacks.norm_scales(net2))
    assert float(model2.compute_ber(net2)) == float(np.float32(zeroed) / np.float32(448))
    # p-value of tools/phash_pvalue.py:34-37: identical hashes -> 2^-256, complementary -> 1
    h = np.random.default_rng(0).integers(0, 2, (3, 256)).astype(bool)
    p = attacks.matching_p_value(h, np.stack([h[0], ~h[1], h[2] ^ (np.arange(256) < 128)]))
    assert p[0] < 1e-30 and p[1] == 1.0 and abs(float(p[2]) - 0.5249) < 1e-3


def test_product_bbox_transforms_match_reference(golden):
    """iprgan.tools trigger transforms (host-side torch ops) against the reference's own classes."""
    from iprgan import Config, tools
    from test_oracle_golden import compare
    compare(cases.run_bbox_transforms(tools, Config), golden('bbox_transforms'), rtol=1e-6, atol=1e-7)


def test_paste_watermark_and_noise_patch(tmp_path):
    """PasteWatermark / RandomNoisePatch (parity unpinned: torchvision + the reference's PNGs are absent):
    product vs oracle restatement on a synthetic RGBA logo, all four corners, opaque and transparent."""
    from PIL import Image
    from iprgan import Config, tools
    from oracle import bbox
    rgba = np.zeros((24, 24, 4), dtype=np.uint8)
    rgba[4:20, 6:18] = (200, 30, 60, 255)
    rgba[8:12, 8:12, 3] = 128
    path = str(tmp_path / 'logo.png')
    Image.fromarray(rgba, 'RGBA').save(path)
    x = torch.tanh(recipe.tensor(5, 1, (2, 3, 32, 32)))
    for pos in ('tl', 'tr', 'bl', 'br'):
        for opaque in (True, False):
            cfg = {'size': 16, 'opaque': opaque, 'watermark': path, 'position': pos}
            a = tools.PasteWatermark(Config(cfg), normalized=True)
            b = bbox.PasteWatermark(gan.Cfg(cfg), normalized=True)
            assert torch.equal(a.fg, b.fg) and torch.equal(a.bg, b.bg)
            assert torch.equal(a(x), b(x))
            assert not torch.equal(a(x), x)
            assert torch.equal(a.apply_mask(x) * 0 + 1, torch.ones(2, 3, 16, 16))
        torch.manual_seed(3)
        a = tools.RandomNoisePatch(Config({'size': 8, 'position': pos}), normalized=False)
        torch.manual_seed(3)
        b = bbox.RandomNoisePatch(gan.Cfg({'size': 8, 'position': pos}), normalized=False)
        assert torch.equal(a(x), b(x)) and list(a.state_dict()) == ['bg', 'fg']
