"""GPU parity of every C-ABI kernel against plain PyTorch fp32 CPU ops (the arithmetic the reference
reaches through torch.nn).  Tolerances: fp32 MFMA is an exact fmaf chain, only the summation order
differs from the CPU, so conv results are compared at 2e-4 of the tensor's max magnitude; integer
results (bit-error counts) must be exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import conftest

pytestmark = pytest.mark.gpu
# the convolution tests run in both math modes ('fp32' and 'fp32x3'), same tolerances (conftest.both_math_modes)
pytest_generate_tests = conftest.both_math_modes({
    'test_conv_fwd_bwd', 'test_conv_dgrad_fused_prev_act', 'test_conv_sn_scale', 'test_reflect_pad_conv',
    'test_every_gconv_tile_variant', 'test_every_wgrad_candidate', 'test_north_star_conv_shapes_full_size',
    'test_conv_epilogue_column_sums', 'test_fused_norm_statistics_with_large_mean_channels', 'test_conv_splitk',
    'test_batchnorm', 'test_instance_norm', 'test_deferred_wgrad_reduce_is_bit_identical',
    'test_reflect_pad_dgrad_without_the_padded_grid'})
math_mode = conftest.math_mode_fixture()


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return torch.device('cuda:0')


def to_nhwc(x):          # CPU helper: NCHW -> NHWC with channel padding to 4
    B, C, H, W = x.shape
    out = torch.zeros(B, H, W, (C + 3) & ~3)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out.contiguous()


def from_nhwc(y, C):
    return y[..., :C].permute(0, 3, 1, 2).contiguous()


def close(a, b, tol=2e-4, what=''):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    scale = max(1e-6, float(b.abs().max()))
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f'{what}: rel-to-max err {err:.3e} > {tol}'


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def test_layout_roundtrip(dev):
    from iprgan import ops
    x = rnd(3, 3, 9, 7)
    y = ops.nchw_to_nhwc(x.to(dev))
    assert torch.equal(y.cpu(), to_nhwc(x))
    assert torch.equal(ops.nhwc_to_nchw(y, 3).cpu(), x)
    x = rnd(2, 20, 5, 6, seed=1)
    assert torch.equal(ops.nhwc_to_nchw(ops.nchw_to_nhwc(x.to(dev)), 20).cpu(), x)
    src = rnd(6 * 5 * 4, seed=2)
    p = ops.permute_021(src.to(dev), 6, 5, 4).cpu()
    assert torch.equal(p, src.view(6, 5, 4).permute(1, 0, 2).reshape(-1))


CONVS = [
    # cin, cout, k, stride, pad, outpad, transposed, H, W, B, act
    (64, 64, 3, 1, 1, 0, False, 16, 16, 2, 'lrelu'),
    (64, 128, 4, 2, 1, 0, False, 16, 16, 3, 'lrelu'),
    (128, 256, 3, 1, 1, 0, False, 8, 8, 4, 'none'),
    (3, 64, 3, 1, 1, 0, False, 16, 16, 2, 'lrelu'),
    (32, 48, 3, 1, 1, 0, False, 7, 9, 2, 'relu'),          # ragged sizes, Cout not a tile multiple
    (64, 32, 3, 2, 1, 0, False, 15, 13, 2, 'none'),        # odd spatial, stride 2
    (128, 64, 4, 2, 1, 0, True, 8, 8, 2, 'none'),          # DCGAN G up-conv
    (512, 256, 4, 2, 1, 0, True, 4, 4, 2, 'none'),
    (64, 3, 3, 1, 1, 0, True, 16, 16, 2, 'tanh'),          # DCGAN G head
    (64, 32, 3, 2, 1, 1, True, 6, 5, 2, 'relu'),           # CycleGAN up-conv (output_padding)
    (256, 512, 3, 1, 1, 0, False, 8, 8, 2, 'lrelu'),
    (128, 128, 1, 1, 0, 0, False, 5, 5, 2, 'none'),        # 1x1
    (512, 64, 6, 1, 0, 0, False, 6, 6, 3, 'lrelu'),        # D96 "FC" conv: full-map path (split-K forward, remapped dgrad)
    (512, 1024, 6, 1, 0, 0, False, 6, 6, 64, 'lrelu'),     # ... at its real size (batch 64, 75 MB of weights)
    (64, 96, 4, 1, 0, 0, False, 4, 4, 70, 'none'),         # full-map, ragged batch (two 64-row tiles) and Cout
    # few output channels, stride 1: backward-weight runs with swapped roles (P = x, Q = dy, negated taps)
    (64, 3, 9, 1, 4, 0, False, 12, 12, 2, 'none'),         # SRGAN head
    (64, 3, 7, 1, 3, 0, False, 13, 11, 2, 'tanh'),         # ragged
    (512, 1, 4, 1, 1, 0, False, 9, 9, 2, 'none'),          # PatchGAN logits
    (64, 16, 3, 1, 1, 0, False, 10, 10, 3, 'none'),
    (1024, 1, 1, 1, 0, 0, False, 1, 1, 5, 'none'),         # D96 last 1x1 on a 1x1 map
]
ACT = {'none': (0, 0.0), 'relu': (1, 0.0), 'lrelu': (2, 0.1), 'tanh': (3, 0.0)}


def ref_act(y, name):
    return {'none': lambda t: t, 'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, 0.1),
            'tanh': torch.tanh}[name](y)


@pytest.mark.parametrize('cfg', CONVS, ids=lambda c: '-'.join(map(str, c)))
def test_conv_fwd_bwd(dev, cfg):
    from iprgan import ops
    cin, cout, k, s, p, op, tr, H, W, B, act = cfg
    x = rnd(B, cin, H, W, seed=1)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = rnd(*wshape, seed=2, scale=(cin * k * k) ** -0.5)
    b = rnd(cout, seed=3, scale=0.1)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    if tr:
        pre = F.conv_transpose2d(xr, wr, br, stride=s, padding=p, output_padding=op)
    else:
        pre = F.conv2d(xr, wr, br, stride=s, padding=p)
    yr = ref_act(pre, act)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)

    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr, act=ACT[act][0], slope=ACT[act][1])
    d = spec.desc(B, H, W)
    xd, wd, bd = to_nhwc(x).to(dev), w.to(dev), b.to(dev)
    wf, wb = ops.conv_prep(spec, d, wd, None, fwd=True, bwd=True)
    y = ops.conv_fwd(spec, d, xd, wf, bd)
    close(from_nhwc(y.cpu(), cout), yr, what='fwd')
    if cout % 4:
        assert float(y[..., cout:].abs().max()) == 0.0, 'pad channels must stay zero'
    gd = to_nhwc(g).to(dev)
    dz = ops.act_bwd(gd, y, *ACT[act]) if act != 'none' else gd
    dw, db = ops.conv_bwd_weight(spec, d, xd, dz, wshape, True)
    close(dw, wr.grad, what='wgrad')
    close(db, br.grad, what='bgrad')
    dx = ops.conv_bwd_data(spec, d, dz, wb)
    close(from_nhwc(dx.cpu(), cin), xr.grad, what='dgrad')


def test_conv_dgrad_fused_prev_act(dev):
    from iprgan import ops
    cin, cout, k = 64, 64, 3
    x_pre = rnd(2, cin, 8, 8, seed=1)
    x = F.leaky_relu(x_pre, 0.1).requires_grad_()
    w = rnd(cout, cin, k, k, seed=2, scale=0.05)
    y = F.conv2d(x, w, None, padding=1)
    g = rnd(*y.shape, seed=3)
    y.backward(g)
    want = x.grad * torch.where(x > 0, 1.0, 0.1)
    spec = ops.ConvSpec(cin, cout, k, 1, 1)
    d = spec.desc(2, 8, 8)
    _, wb = ops.conv_prep(spec, d, w.to(dev), None, fwd=False, bwd=True)
    xd = to_nhwc(x.detach()).to(dev)
    dx = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb, xd, 2, 0.1)
    close(from_nhwc(dx.cpu(), cin), want, what='fused dgrad')


def test_conv_sn_scale(dev):
    from iprgan import ops
    w = rnd(64, 64, 3, 3, seed=5, scale=0.05)
    x = rnd(2, 64, 8, 8, seed=6)
    sigma = torch.tensor([1.7])
    spec = ops.ConvSpec(64, 64, 3, 1, 1)
    d = spec.desc(2, 8, 8)
    wf, _ = ops.conv_prep(spec, d, w.to(dev), sigma.to(dev))
    y = ops.conv_fwd(spec, d, to_nhwc(x).to(dev), wf, None)
    close(from_nhwc(y.cpu(), 64), F.conv2d(x, w / sigma, padding=1), what='sn-scaled conv')


@pytest.mark.parametrize('M,C,act', [(2 * 16 * 16, 64, 'relu'), (777, 128, 'lrelu'), (5000, 256, 'none'), (3, 64, 'relu'),
                                     (1234, 96, 'relu'), (300, 20, 'lrelu'), (4097, 512, 'none')])   # C4n not dividing 256: generic apply path
def test_batchnorm(dev, M, C, act):
    from iprgan import ops
    x = rnd(M, C, seed=1) * 2 + 0.5
    gamma, beta = rnd(C, seed=2) * 0.5 + 1, rnd(C, seed=3) * 0.1
    rm, rv = rnd(C, seed=4) * 0.1, torch.rand(C) + 0.5
    bn = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm); bn.running_var.copy_(rv)
    xr = x.clone().view(M, C, 1, 1).requires_grad_()
    yr = ref_act(bn(xr), act)
    g = rnd(M, C, seed=5)
    yr.backward(g.view(M, C, 1, 1))
    rmd, rvd = rm.to(dev), rv.to(dev)
    xd = x.to(dev).view(1, M, 1, C)
    y, mean, invstd = ops.bn_fwd(xd, gamma.to(dev), beta.to(dev), rmd, rvd, 1e-5, 0.1, True, *ACT[act])
    close(y.view(M, C), yr.view(M, C), 1e-4, 'bn fwd')
    close(rmd, bn.running_mean, 1e-5, 'running_mean')
    close(rvd, bn.running_var, 1e-5, 'running_var')
    # y is passed as None for relu / lrelu / none: the derivative mask is recomputed from x (needs gamma AND beta);
    # dbias = column sums of dx (bias gradient of a producing conv) accumulated on the apply pass
    dbias = torch.full((C,), 7.0, device=dev)
    dx, dg, db = ops.bn_bwd(xd, None, g.to(dev).view(1, M, 1, C), gamma.to(dev), mean, invstd, *ACT[act],
                            beta=beta.to(dev), dbias=dbias, dbias_beta=1.0)
    close(dx.view(M, C), xr.grad.view(M, C), 2e-4, 'bn dx')
    want_db = xr.grad.view(M, C).double().sum(0)
    assert float((dbias.cpu().double() - 7.0 - want_db).abs().max()) <= 1e-3 * float(xr.grad.abs().max()) * M ** 0.5
    close(dg, bn.weight.grad, 2e-4, 'bn dgamma')
    close(db, bn.bias.grad, 2e-4, 'bn dbeta')
    # eval mode uses running stats
    bn.eval()
    ye = ref_act(bn(x.view(M, C, 1, 1)), act)
    y2, _, _ = ops.bn_fwd(xd, gamma.to(dev), beta.to(dev), rmd, rvd, 1e-5, 0.1, False, *ACT[act])
    close(y2.view(M, C), ye.view(M, C), 1e-4, 'bn eval')


BN_DGRAD_CASES = [  # cin (norm channels), cout, k, s, p, transposed, H, W, B, act
    (64, 128, 3, 1, 1, False, 13, 11, 5, 'relu'),        # ragged M, two 64-row tile rows of the small tiles
    (128, 64, 4, 2, 1, False, 16, 16, 4, 'lrelu'),       # strided consumer: four sub-pixel phases write dz
    (64, 32, 4, 2, 1, True, 8, 8, 6, 'relu'),            # ConvTranspose consumer (the generator's case)
    (256, 256, 3, 1, 1, False, 8, 8, 8, 'none'),         # no activation behind the norm; 256-wide tiles apply
]


@pytest.mark.parametrize('mode', ['fp32', 'bf16act'])
@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 4, 5, 8, 9, 10, 11, 13, 14, 16, 17])
@pytest.mark.parametrize('case', BN_DGRAD_CASES, ids=lambda c: '-'.join(map(str, c)))
def test_conv_bwd_data_into_batchnorm(dev, case, tile, mode):
    """iprgan_conv_bwd_data_bn + iprgan_bn_bwd_pre (x -> BatchNorm -> act -> conv): the consumer's backward-data epilogue
    applies the activation derivative (mask recomputed from x with the forward's expression) and takes the two reductions
    of the norm backward; the norm backward then needs no reduction pass.  Against torch autograd for dx, dgamma, dbeta
    and the bias-gradient column sums, through every tile family (a tile that does not apply falls back)."""
    from iprgan import _lib, ops
    cin, cout, k, s_, p, tr, H, W, B, actname = case
    act = {'relu': 1, 'lrelu': 2, 'none': 0}[actname]
    slope = 0.2 if actname == 'lrelu' else 0.0
    x = rnd(B, cin, H, W, seed=1)
    gamma, beta = rnd(cin, seed=2, scale=0.5) + 1.0, rnd(cin, seed=3, scale=0.3)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = rnd(*wshape, seed=4, scale=(cin * k * k) ** -0.5)
    if mode == 'bf16act':
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xr, gr, br = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    bn = F.batch_norm(xr, None, None, gr, br, training=True, eps=1e-5)
    a = {'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, 0.2), 'none': lambda t: t}[actname](bn)
    y = F.conv_transpose2d(a, w, None, stride=s_, padding=p) if tr else F.conv2d(a, w, None, stride=s_, padding=p)
    gy = rnd(*y.shape, seed=5)
    if mode == 'bf16act':
        gy = gy.bfloat16().float()
    y.backward(gy)
    try:
        _lib.set_math(mode)
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        xd = to_nhwc(x).to(dev)
        adt = ops.act_dtype(cin)
        xd = xd.to(adt)
        yb, mean, invstd = ops.bn_fwd(xd, gamma.to(dev), beta.to(dev), None, None, 1e-5, 0.0, True, act, slope)
        spec = ops.ConvSpec(cin, cout, k, s_, p, 0, tr)
        d = spec.desc(B, H, W)
        assert ops.conv_bwd_data_bn_ok(d)
        _, wb = ops.conv_prep(spec, d, w.to(dev), None, False, True)
        gd = to_nhwc(gy).to(dev).to(ops.act_dtype(cout))
        dz, partials = ops.conv_bwd_data_bn(spec, d, gd, wb, xd, mean, invstd, gamma.to(dev), beta.to(dev), act, slope)
        dbias = torch.zeros(cin, device=dev)
        dx, dg, db = ops.bn_bwd_pre(xd, dz, gamma.to(dev), mean, invstd, partials, dbias=dbias)
        tol = 2e-4 if mode == 'fp32' else 2e-2
        close(from_nhwc(dx.float().cpu(), cin), xr.grad, tol, f'dx tile {tile}')
        close(dg.cpu(), gr.grad, tol, f'dgamma tile {tile}')
        close(db.cpu(), br.grad, tol, f'dbeta tile {tile}')
        # column sums of dx (the bias gradient of a convolution below): sum over positions of a batch-norm gradient is 0
        assert float(dbias.abs().max()) <= (1e-3 if mode == 'fp32' else 5e-2) * float(xr.grad.abs().max()) * (B * H * W) ** 0.5
        # and the unfused pair of calls gives the same result (same mask rule)
        dy_plain = ops.conv_bwd_data(spec, d, gd, wb)
        dx2, dg2, db2 = ops.bn_bwd(xd, yb, dy_plain, gamma.to(dev), mean, invstd, act, slope, beta=beta.to(dev))
        close(dx.float(), dx2.float(), tol, 'fused vs unfused dx')
        close(dg, dg2, tol, 'fused vs unfused dgamma')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


@pytest.mark.parametrize('shape', [(2, 5, 7, 64), (3, 12, 12, 8), (1, 1, 1, 4)])
def test_pixel_shuffle_with_fused_prelu(shape, dev):
    """iprgan_pixel_shuffle2_prelu_fwd / _bwd against torch's pixel_shuffle + prelu (sr_resnet.py:39-45), and against the
    two separate kernels bit for bit (the same arithmetic, one pass)."""
    import torch.nn.functional as F
    from iprgan import ops
    B, H, W, C = shape
    g = torch.Generator().manual_seed(B * 100 + C)
    x = torch.randn(B, H, W, 4 * C, generator=g).to(dev)
    dy = torch.randn(B, 2 * H, 2 * W, C, generator=g).to(dev)
    alpha = torch.tensor([0.23], device=dev)
    y = ops.pixel_shuffle2_prelu_fwd(x, alpha)
    dx, da = ops.pixel_shuffle2_prelu_bwd(x, dy, alpha)
    xt = x.permute(0, 3, 1, 2).detach().clone().requires_grad_(True)
    at = alpha.detach().clone().requires_grad_(True)
    yt = F.prelu(F.pixel_shuffle(xt, 2), at)
    yt.backward(dy.permute(0, 3, 1, 2))
    assert torch.equal(y, yt.detach().permute(0, 2, 3, 1).contiguous())
    assert torch.equal(dx, xt.grad.permute(0, 2, 3, 1).contiguous())
    assert abs(float(da) - float(at.grad)) <= 2e-6 * max(1.0, float((dy.abs() * F.pixel_shuffle(xt, 2).permute(0, 2, 3, 1).abs()).sum()) ** 0.5)
    s = ops.pixel_shuffle2(x)
    y2 = ops.prelu_fwd(s, alpha)
    d2, da2 = ops.prelu_bwd(s, dy, alpha)
    assert torch.equal(y, y2) and torch.equal(dx, ops.pixel_shuffle2(d2, inverse=True))
    assert abs(float(da) - float(da2)) <= 1e-5 * max(1.0, abs(float(da2)))


@pytest.mark.parametrize('M,C', [(2 * 24 * 24, 64), (777, 64), (5 * 12 * 12, 256)])
def test_batchnorm_with_fused_prelu(dev, M, C):
    """iprgan_bn_prelu_fwd / _bwd (networks/sr_resnet.py:7,13: conv -> BatchNorm -> PReLU with one learnable slope read
    from the device): output, dx, dgamma, dbeta, the slope gradient and the bias-gradient column sums against torch."""
    from iprgan import ops
    x = rnd(M, C, seed=1) * 1.5 + 0.3
    gamma, beta = rnd(C, seed=2, scale=0.5) + 1.0, rnd(C, seed=3, scale=0.3)
    alpha = torch.tensor([0.25])
    xr, gr, br, ar = x.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_(), alpha.clone().requires_grad_()
    y_ref = F.prelu(F.batch_norm(xr.view(M, C, 1, 1), None, None, gr, br, training=True, eps=1e-5), ar).view(M, C)
    g = rnd(M, C, seed=4)
    y_ref.backward(g)
    xd, ad = x.to(dev), alpha.to(dev)
    y, mean, invstd = ops.bn_prelu_fwd(xd, gamma.to(dev), beta.to(dev), None, None, 1e-5, 0.0, True, ad)
    close(y, y_ref, 1e-5, 'bn+prelu fwd')
    dbias = torch.zeros(C, device=dev)
    dx, dg, db, da = ops.bn_prelu_bwd(xd, g.to(dev), gamma.to(dev), beta.to(dev), mean, invstd, ad, dbias=dbias)
    close(dx, xr.grad, 2e-4, 'bn+prelu dx')
    close(dg, gr.grad, 2e-4, 'dgamma')
    close(db, br.grad, 2e-4, 'dbeta')
    close(da, ar.grad, 2e-4, 'dslope')
    assert float(dbias.abs().max()) <= 1e-3 * float(xr.grad.abs().max()) * M ** 0.5


@pytest.mark.parametrize('rows,cols', [(64, 27), (512, 2304), (1, 32768), (128, 2048)])
def test_spectral_norm(dev, rows, cols):
    from iprgan import ops
    w = rnd(rows, cols, seed=1, scale=cols ** -0.5)
    u = F.normalize(rnd(rows, seed=2), dim=0)
    v = F.normalize(rnd(cols, seed=3), dim=0)
    # torch semantics (nn/utils/spectral_norm.py compute_weight)
    v1 = F.normalize(torch.mv(w.t(), u), dim=0, eps=1e-12)
    u1 = F.normalize(torch.mv(w, v1), dim=0, eps=1e-12)
    wr = w.clone().requires_grad_()
    sig = torch.dot(u1, torch.mv(wr, v1))
    wsn = wr / sig
    g = rnd(rows, cols, seed=4)
    wsn.backward(g)
    ud, vd, wd = u.to(dev), v.to(dev), w.to(dev)
    sigma = ops.sn_power_iter(wd, ud, vd, True)
    close(ud, u1, 1e-5, 'u'); close(vd, v1, 1e-5, 'v'); close(sigma, sig.view(1), 1e-5, 'sigma')
    dw = ops.sn_bwd(g.to(dev), wd, ud, vd, sigma)
    close(dw, wr.grad, 1e-4, 'sn bwd')
    u_before = ud.clone()
    s2 = ops.sn_power_iter(wd, ud, vd, False)           # eval: no update
    assert torch.equal(u_before, ud)
    close(s2, torch.dot(u1, torch.mv(w, v1)).view(1), 1e-5, 'sigma eval')


@pytest.mark.parametrize('shapes', [[(64, 27), (128, 1024), (1, 8192)], [(64, 576), (1, 131072), (8, 40000), (512, 2304)]],
                         ids=['narrow', 'wide'])
def test_spectral_norm_multi(dev, shapes):
    """All layers of a discriminator in one call (iprgan_sn_power_iter_multi); the 'wide' table takes the split form of
    the v normalisation (columns > 32768: the 131072-column Linear head of sn_discriminator.py:27-32 at 128x128)."""
    from iprgan import ops
    ws = [rnd(r, c, seed=10 + i, scale=c ** -0.5) for i, (r, c) in enumerate(shapes)]
    us = [F.normalize(rnd(r, seed=20 + i), dim=0) for i, (r, c) in enumerate(shapes)]
    vs = [F.normalize(rnd(c, seed=30 + i), dim=0) for i, (r, c) in enumerate(shapes)]
    wd, ud, vd = [w.to(dev) for w in ws], [u.to(dev) for u in us], [v.to(dev) for v in vs]
    sig, u_out, v_out = ops.sn_power_iter_multi(wd, ud, vd, True)
    for i, (w, u, v) in enumerate(zip(ws, us, vs)):
        v1 = F.normalize(torch.mv(w.t(), u), dim=0, eps=1e-12)
        u1 = F.normalize(torch.mv(w, v1), dim=0, eps=1e-12)
        close(vd[i], v1, 1e-5, f'v[{i}]'); close(ud[i], u1, 1e-5, f'u[{i}]')
        assert torch.equal(v_out[i], vd[i]) and torch.equal(u_out[i], ud[i])
        close(sig[i:i + 1], torch.dot(u1, torch.mv(w, v1)).view(1), 1e-5, f'sigma[{i}]')
    sig2, _, _ = ops.sn_power_iter_multi(wd, [u.to(dev) for u in us], [v.to(dev) for v in vs], True)
    assert torch.equal(sig, sig2)              # fixed-order partial sums: bit-identical on repeat


def test_gemv_head(dev):
    from iprgan import ops
    B, K = 5, 2048
    x, w = rnd(B, K, seed=1), rnd(K, seed=2, scale=0.02)
    b, sigma, g = torch.tensor([0.3]), torch.tensor([1.3]), rnd(B, seed=3)
    y = ops.gemv_fwd(x.to(dev), w.to(dev), b.to(dev), sigma.to(dev))
    close(y, x @ (w / sigma) + b, 1e-5, 'gemv fwd')
    dx, dw, db = ops.gemv_bwd(x.to(dev), w.to(dev), g.to(dev), sigma.to(dev), True, True)
    close(dx, g[:, None] * (w / sigma)[None], 1e-5, 'gemv dx')
    close(dw, g @ x, 1e-5, 'gemv dw')
    close(db, g.sum().view(1), 1e-5, 'gemv db')
    # backward-weight: 16-byte loads (4 fp32 / 8 bf16 columns per thread), more samples than one unrolled round,
    # ragged K (scalar tail columns), bf16 activations
    for B2, K2, b16 in ((19, 2048, False), (19, 2050, False), (21, 4096, True), (9, 1028, True)):
        x2 = rnd(B2, K2, seed=7)
        g2 = rnd(B2, seed=8)
        xd = x2.to(dev).bfloat16() if b16 else x2.to(dev)
        xr = x2.bfloat16().float() if b16 else x2
        w2 = rnd(K2, seed=9, scale=0.02).to(dev)
        _, dw2, db2 = ops.gemv_bwd(xd, w2, g2.to(dev), sigma.to(dev), False, True)
        close(dw2.float(), g2 @ xr, 1e-5, f'gemv dw B{B2} K{K2} bf16={b16}')
        close(db2.float(), g2.sum().view(1), 1e-5, 'gemv db')


LOSSES = {
    0: lambda x, y: F.relu(1 - x).mean(), 1: lambda x, y: F.relu(1 + x).mean(), 2: lambda x, y: -x.mean(),
    3: lambda x, y: F.binary_cross_entropy_with_logits(x, torch.ones_like(x)),
    4: lambda x, y: F.binary_cross_entropy_with_logits(x, torch.zeros_like(x)),
    5: lambda x, y: F.mse_loss(x, torch.ones_like(x)), 6: lambda x, y: F.mse_loss(x, torch.zeros_like(x)),
    7: F.mse_loss, 8: F.l1_loss,
}


@pytest.mark.parametrize('kind', list(LOSSES))
@pytest.mark.parametrize('n', [7, 128, 100003])
def test_losses(dev, kind, n):
    from iprgan import tools
    x = (rnd(n, seed=1) * 2).requires_grad_()
    y = rnd(n, seed=2)
    ref = LOSSES[kind](x, y) * 1.7
    ref.backward()
    xd = x.detach().to(dev).requires_grad_()
    out = tools.loss_value(kind, xd, y.to(dev) if kind >= 7 else None) * 1.7
    out.backward()
    close(out, ref, 1e-5, 'loss')
    close(xd.grad, x.grad, 1e-5, 'loss grad')


@pytest.mark.parametrize('n', [5, 4 * 3 * 32 * 32])
def test_vae_losses_and_reparam(dev, n):
    """models/vae.py:36-48 (KL and BCE, sum / N with ATen's log clamp) and networks/encoder.py:24-28."""
    from iprgan import _lib as L, ops, tools
    N = 4
    x = torch.tanh(rnd(n, seed=1) * 3)
    x[:2] = torch.tensor([1.0, -1.0])[:min(2, n)]          # p = 1 and p = 0: the clamped-log / 1e-12 branches
    x = x.requires_grad_()
    t = torch.tanh(rnd(n, seed=2))
    ref = F.binary_cross_entropy((x + 1.) / 2., (t + 1.) / 2., reduction='sum') / N
    (ref * 0.7).backward()
    xd = x.detach().to(dev).requires_grad_()
    out = tools.loss_sum(L.LOSS_BCE_PM1, xd, t.to(dev), 1.0 / N)
    (out * 0.7).backward()
    close(out, ref, 1e-5, 'bce')
    # the gradient at exactly p in {0, 1} is (p - t) / 1e-12 in ATen and here: compare the finite entries tightly
    close(xd.grad[2:], x.grad[2:], 1e-5, 'bce grad')
    assert torch.equal(torch.isfinite(xd.grad.cpu()), torch.isfinite(x.grad))
    m, lv = rnd(n, seed=3).requires_grad_(), (rnd(n, seed=4) * 0.5).requires_grad_()
    ref = ((m ** 2 + lv.exp() - 1 - lv) / 2).sum() / N
    ref.backward()
    md, lvd = m.detach().to(dev).requires_grad_(), lv.detach().to(dev).requires_grad_()
    out = tools.loss_sum(L.LOSS_KL_MEAN, md, None, 1.0 / N) + tools.loss_sum(L.LOSS_KL_LOGVAR, lvd, None, 1.0 / N)
    out.backward()
    close(out, ref, 1e-5, 'kl')
    close(md.grad, m.grad, 1e-6, 'kl dmean')
    close(lvd.grad, lv.grad, 1e-5, 'kl dlogvar')
    eps, g = rnd(n, seed=5), rnd(n, seed=6)
    m2, lv2 = m.detach().clone().requires_grad_(), lv.detach().clone().requires_grad_()
    z = eps * torch.exp(lv2 * 0.5) + m2
    z.backward(g)
    zd = ops.reparam_fwd(m.detach().to(dev), lv.detach().to(dev), eps.to(dev))
    dm, dlv = ops.reparam_bwd(g.to(dev), lv.detach().to(dev), eps.to(dev))
    close(zd, z, 1e-6, 'reparam')
    close(dm, m2.grad, 1e-6, 'reparam dmean')
    close(dlv, lv2.grad, 1e-6, 'reparam dlogvar')


@pytest.mark.parametrize('shape', [(2, 3, 32, 32), (1, 3, 11, 11), (3, 1, 37, 53), (2, 3, 64, 64)])
@pytest.mark.parametrize('denorm', [False, True])
def test_ssim_loss_vs_oracle(dev, shape, denorm):
    """iprgan_ssim_fwd/bwd against the CPU restatement of pytorch-msssim's ssim (oracle/ssim.py, parity unpinned):
    loss to 1e-5, gradient to 2e-4 of its max; ragged sizes cover partial tiles; SSIM(x, x) = 1."""
    from iprgan import tools
    from oracle import ssim as ossim
    g = torch.Generator().manual_seed(sum(shape))
    x, y = torch.rand(*shape, generator=g), torch.rand(*shape, generator=g)
    y = 0.7 * y + 0.3 * x                                  # correlated, like a generated image and its target
    if denorm:
        x, y = x * 2 - 1, y * 2 - 1
    xr = x.clone().requires_grad_()
    ref = ossim.ssim_loss(normalized=denorm)(xr, y)
    (ref * 1.3).backward()
    xd = x.to(dev).requires_grad_()
    out = tools.ssim(normalized=denorm)(xd, y.to(dev))
    (out * 1.3).backward()
    assert abs(float(out.detach()) - float(ref.detach())) < 1e-5, (float(out.detach()), float(ref.detach()))
    close(xd.grad, xr.grad, 2e-4, 'ssim grad')
    same = tools.ssim(normalized=denorm)(x.to(dev), x.to(dev))
    assert abs(float(same)) < 1e-6


def test_sign_loss_and_ber_exact(dev):
    from iprgan import ops
    g = np.random.default_rng(5)
    sizes = [256, 128, 64, 7]
    gammas = [torch.from_numpy(g.standard_normal(n).astype(np.float32) * 0.3) for n in sizes]
    gammas[0][:5] = 0.0                                   # sign(0) = 0 counts as an error
    signs = [torch.from_numpy((g.integers(0, 2, n) * 2 - 1).astype(np.float32)) for n in sizes]
    gr = [t.clone().requires_grad_() for t in gammas]
    ref = sum(F.relu(0.1 - a * b).mean() for a, b in zip(gr, signs))
    (ref * 2.5).backward()
    gd, sd = [t.to(dev) for t in gammas], [t.to(dev) for t in signs]
    loss = ops.sign_loss_fwd(gd, sd, 0.1)
    close(loss, ref, 1e-6, 'sign loss')
    grads = ops.sign_loss_bwd(gd, sd, 0.1, torch.tensor(2.5, device=dev))
    for a, b in zip(grads, gr):
        close(a, b.grad, 1e-6, 'sign grad')
    counts = ops.sign_ber_counts(gd, sd).cpu()
    wrong = sum(int((a.sign() != b).sum()) for a, b in zip(gammas, signs))
    assert counts.tolist() == [wrong, sum(sizes)]


def test_adam_matches_torch(dev):
    from iprgan import optim
    shapes = [(64, 3, 3, 3), (64,), (1000, 17), (1,)] * 20        # > one launch table
    ps = [rnd(*s, seed=i) for i, s in enumerate(shapes)]
    ref = [p.clone().requires_grad_() for p in ps]
    mine = [p.clone().to(dev).requires_grad_() for p in ps]
    o1 = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
    o2 = optim.Adam(mine, lr=2e-4, betas=(0.5, 0.999))
    for it in range(3):
        for i, (a, b) in enumerate(zip(ref, mine)):
            g = rnd(*a.shape, seed=100 * it + i)
            a.grad, b.grad = g.clone(), g.to(dev)
        o1.step(); o2.step()
    for a, b in zip(ref, mine):
        close(b, a, 1e-6, 'adam param')
    sd1, sd2 = o1.state_dict(), o2.state_dict()
    assert sd1['param_groups'][0]['betas'] == sd2['param_groups'][0]['betas']
    assert set(sd2['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    close(sd2['state'][2]['exp_avg_sq'], sd1['state'][2]['exp_avg_sq'], 1e-6, 'exp_avg_sq')
    assert float(sd2['state'][0]['step']) == 3.0


def test_cpu_tensor_is_refused(dev):
    from iprgan import networks, ops
    with pytest.raises(RuntimeError):
        ops.loss_fwd(0, torch.zeros(4))
    with pytest.raises(RuntimeError):
        networks.ConvGenerator32()(torch.zeros(2, 128))


REFLECT_CONVS = [(64, 64, 3, 1, 1, 8, 9, 2), (3, 32, 7, 1, 3, 12, 10, 2), (64, 3, 7, 1, 3, 9, 9, 1), (256, 256, 3, 1, 1, 6, 6, 2),
                 (64, 3, 7, 1, 3, 14, 11, 2), (3, 64, 7, 1, 3, 13, 16, 2)]         # RGB head, ragged: swapped-role wgrad on the reflect-padded copy


@pytest.mark.parametrize('cfg', REFLECT_CONVS, ids=lambda c: '-'.join(map(str, c)))
def test_reflect_pad_conv(dev, cfg):
    """ReflectionPad2d(p) + Conv2d(k, pad=0) folded into the gather (resnet_generator.py:6-7,41-49)."""
    from iprgan import ops
    cin, cout, k, s, p, H, W, B = cfg
    x, w, b = rnd(B, cin, H, W, seed=1), rnd(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5), rnd(cout, seed=3)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.conv2d(F.pad(xr, (p, p, p, p), mode='reflect'), wr, br, stride=s)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)
    spec = ops.ConvSpec(cin, cout, k, s, p, pad_mode=1)
    d = spec.desc(B, H, W)
    xd = to_nhwc(x).to(dev)
    wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
    y = ops.conv_fwd(spec, d, xd, wf, b.to(dev))
    close(from_nhwc(y.cpu(), cout), yr, what='reflect fwd')
    gd = to_nhwc(g).to(dev)
    dw, db = ops.conv_bwd_weight(spec, d, xd, gd, w.shape, True)
    close(dw, wr.grad, what='reflect wgrad'); close(db, br.grad, what='reflect bgrad')
    close(from_nhwc(ops.conv_bwd_data(spec, d, gd, wb).cpu(), cin), xr.grad, what='reflect dgrad')


@pytest.mark.parametrize('affine,act', [(True, 'relu'), (False, 'lrelu'), (True, 'none')])
def test_instance_norm(dev, affine, act):
    from iprgan import ops
    B, C, H, W = 3, 64, 9, 7
    x = rnd(B, C, H, W, seed=1) * 2 + 0.3
    m = torch.nn.InstanceNorm2d(C, affine=affine)
    if affine:
        with torch.no_grad():
            m.weight.copy_(rnd(C, seed=2) * 0.5 + 1); m.bias.copy_(rnd(C, seed=3) * 0.1)
    xr = x.clone().requires_grad_()
    yr = {'none': lambda t: t, 'relu': F.relu, 'lrelu': lambda t: F.leaky_relu(t, 0.1)}[act](m(xr))
    g = rnd(B, C, H, W, seed=4)
    yr.backward(g)
    gam = m.weight.detach().to(dev) if affine else None
    bet = m.bias.detach().to(dev) if affine else None
    xd = to_nhwc(x).to(dev)
    y, mean, invstd = ops.instnorm_fwd(xd, gam, bet, 1e-5, *ACT[act])
    close(from_nhwc(y.cpu(), C), yr, 1e-4, 'in fwd')
    dbias = torch.zeros(C, device=dev)
    dx, dg, db = ops.instnorm_bwd(xd, None, to_nhwc(g).to(dev), gam, mean, invstd, *ACT[act], beta=bet, dbias=dbias)
    close(from_nhwc(dx.cpu(), C), xr.grad, 2e-4, 'in dx')
    want_db = xr.grad.double().sum((0, 2, 3))
    assert float((dbias.cpu().double() - want_db).abs().max()) <= 1e-3 * float(xr.grad.abs().max()) * (B * H * W) ** 0.5
    if affine:
        close(dg, m.weight.grad, 2e-4, 'in dgamma'); close(db, m.bias.grad, 2e-4, 'in dbeta')


def test_prelu_pixelshuffle_maxpool_add(dev):
    from iprgan import ops
    x = rnd(2, 16, 6, 8, seed=1)
    a = torch.tensor([0.25])
    xr, ar = x.clone().requires_grad_(), a.clone().requires_grad_()
    yr = F.prelu(xr, ar)
    g = rnd(*x.shape, seed=2)
    yr.backward(g)
    xd = to_nhwc(x).to(dev)
    close(from_nhwc(ops.prelu_fwd(xd, a.to(dev)).cpu(), 16), yr, 1e-6, 'prelu')
    dx, da = ops.prelu_bwd(xd, to_nhwc(g).to(dev), a.to(dev))
    close(from_nhwc(dx.cpu(), 16), xr.grad, 1e-6, 'prelu dx'); close(da, ar.grad, 1e-5, 'prelu dalpha')
    ps = F.pixel_shuffle(x, 2)
    y = ops.pixel_shuffle2(xd)
    assert torch.equal(from_nhwc(y.cpu(), 4), ps)
    assert torch.equal(ops.pixel_shuffle2(y, inverse=True).cpu(), xd.cpu())
    xr2 = x.clone().requires_grad_()
    mp = F.max_pool2d(xr2, 2, 2)
    g2 = rnd(*mp.shape, seed=3)
    mp.backward(g2)
    assert torch.equal(from_nhwc(ops.maxpool2_fwd(xd).cpu(), 16), mp.detach())
    assert torch.equal(from_nhwc(ops.maxpool2_bwd(xd, to_nhwc(g2).to(dev)).cpu(), 16), xr2.grad)
    assert torch.equal(ops.add(xd, xd).cpu(), (xd + xd).cpu())


TILE_SHAPES = [(256, 512, 3, 1, 1, False, 16, 2), (64, 64, 4, 2, 1, False, 16, 3), (128, 64, 4, 2, 1, True, 8, 2),
               (64, 128, 3, 1, 1, False, 9, 1)]


@pytest.mark.parametrize('tile', list(range(6)) + [8, 9, 10, 11, 12, 13])
def test_every_gconv_tile_variant(dev, tile):
    """The autotuner picks ONE tile per geometry; this forces each of the forward/backward-data tile variants in turn
    (including shapes with few rows) so that none ships untested.  8-13: the LDS-DMA ring tiles of conv_pipe.hip in their
    fp32 form (fp32 operands in HBM, exact fp32 MFMA)."""
    from iprgan import _lib, ops
    try:
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        for cin, cout, k, s, p, tr, H, B in TILE_SHAPES:
            x = rnd(B, cin, H, H, seed=1)
            wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
            w = rnd(*wshape, seed=2, scale=(cin * k * k) ** -0.5)
            xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
            yr = F.conv_transpose2d(xr, wr, None, stride=s, padding=p) if tr else F.conv2d(xr, wr, None, stride=s, padding=p)
            g = rnd(*yr.shape, seed=3)
            yr.backward(g)
            spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
            d = spec.desc(B, H, H)
            wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
            y = ops.conv_fwd(spec, d, to_nhwc(x).to(dev), wf, None)
            close(from_nhwc(y.cpu(), cout), yr, what=f'tile {tile} fwd {cin}->{cout}')
            dx = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb)
            close(from_nhwc(dx.cpu(), cin), xr.grad, what=f'tile {tile} dgrad {cin}->{cout}')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)


@pytest.mark.parametrize('cand', range(60))
def test_every_wgrad_candidate(dev, cand):
    """cand = 20 * variant + 4 * block target + tile shape; variant 0 = [m][n] LDS image, 1 = transposed image
    (wgrad_t_kernel), 2 = transposed image with XCD-contiguous splits.  A candidate that does not apply to a layer falls
    back to the default inside the library, so every shape is checked under every forced candidate."""
    from iprgan import _lib, ops
    try:
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        for cin, cout, k, s, p, tr, H, B in TILE_SHAPES:
            x = rnd(B, cin, H, H, seed=1)
            wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
            w = rnd(*wshape, seed=2, scale=0.05)
            xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
            br = torch.zeros(cout, requires_grad=True)
            yr = F.conv_transpose2d(xr, wr, br, stride=s, padding=p) if tr else F.conv2d(xr, wr, br, stride=s, padding=p)
            g = rnd(*yr.shape, seed=3)
            yr.backward(g)
            spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
            d = spec.desc(B, H, H)
            dw, db = ops.conv_bwd_weight(spec, d, to_nhwc(x).to(dev), to_nhwc(g).to(dev), wshape, True)
            close(dw, wr.grad, what=f'cand {cand} wgrad {cin}->{cout}')
            close(db, br.grad, what=f'cand {cand} bgrad')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)


HALO_SHAPES = [  # cin, cout, k, s, p, outpad, transposed, H, W, B
    (64, 64, 3, 1, 1, 0, False, 16, 16, 3),       # k3 s1: one 64x64 tile pair, zero padding on every border patch
    (64, 128, 3, 1, 1, 0, False, 13, 11, 2),      # ragged maps: patches that hang over the right / bottom edge
    (128, 64, 4, 2, 1, 0, False, 16, 16, 4),      # k4 s2 Conv2d (D.conv1 form): two channel chunks of the gathered tensor
    (64, 64, 4, 2, 1, 0, False, 18, 14, 2),       # k4 s2, ragged output grid 9 x 7
    (128, 64, 4, 2, 1, 0, True, 8, 8, 5),         # ConvT k4 s2 (generator): S = x on the input grid, L = dy
    (64, 128, 4, 2, 1, 0, True, 5, 6, 3),         # ConvT, small ragged grid
]


@pytest.mark.gpu
@pytest.mark.parametrize('cand', [60, 61, 62, 63, 64, 65, 66, 67, 68])
@pytest.mark.parametrize('shape', HALO_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_wgrad_halo(dev, shape, cand):
    """Backward-weight in the halo form (wgrad_halo.hip; candidates 60-68 = 3 variants x block targets 128 / 256 / 512) on
    bf16-representable tensors (products exact, fp32 accumulation): every tap, stride residue, zero-padded border and
    ragged patch against torch, for Conv2d and ConvTranspose2d; accumulate-into-bucket form (beta = 1) as well."""
    from iprgan import _lib, ops
    cin, cout, k, s, p, op, tr, H, W, B = shape
    x = rnd(B, cin, H, W, seed=1).bfloat16().float()
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.zeros(*wshape, requires_grad=True)
    conv = (lambda t: F.conv_transpose2d(t, w, None, stride=s, padding=p, output_padding=op)) if tr else \
           (lambda t: F.conv2d(t, w, None, stride=s, padding=p))
    y = conv(x)
    g = rnd(*y.shape, seed=4).bfloat16().float()
    y.backward(g)
    try:
        _lib.set_math('bf16act')
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
        d = spec.desc(B, H, W)
        xd, gd = to_nhwc(x).to(dev).bfloat16(), to_nhwc(g).to(dev).bfloat16()
        dw, _ = ops.conv_bwd_weight(spec, d, xd, gd, wshape, False)
        close(dw, w.grad, 2e-4, f'halo wgrad cand {cand}')
        acc = torch.full(wshape, 0.5, device=dev)
        ops.conv_bwd_weight(spec, d, xd, gd, wshape, False, dw=acc, beta=1.0)
        close(acc.cpu(), w.grad + 0.5, 2e-4, f'halo wgrad beta=1 cand {cand}')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


HALO32_SHAPES = [  # cin, cout, k, s, p, outpad, transposed, pad_mode, H, W, B
    (64, 64, 3, 1, 1, 0, False, 0, 16, 16, 3),    # k3 s1 zero padding
    (32, 128, 3, 1, 1, 0, False, 1, 13, 11, 2),   # k3 s1 ReflectionPad2d (CycleGAN residual convolutions), ragged maps
    (64, 64, 3, 2, 1, 0, False, 0, 17, 15, 2),    # k3 s2 (Discriminator96 / Resnet down-sampling): four stride residues
    (64, 128, 4, 2, 1, 0, False, 0, 16, 16, 3),   # k4 s2 Conv2d
    (128, 64, 4, 2, 1, 0, True, 0, 8, 8, 4),      # k4 s2 ConvTranspose2d
    (64, 64, 4, 1, 1, 0, False, 0, 12, 10, 2),    # k4 s1 (PatchGAN), output 11 x 9
    (64, 32, 3, 2, 1, 1, True, 0, 6, 5, 2),       # ConvT k3 s2 with output_padding (Resnet up-sampling)
]


@pytest.mark.gpu
@pytest.mark.parametrize('cand', [71, 72, 73])
@pytest.mark.parametrize('shape', HALO32_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_wgrad_halo_fp32(dev, shape, cand):
    """Backward-weight in the halo form on the exact fp32 MFMA (wgrad_halo_f32_kernel; candidates 71-73 = block targets
    256 / 512 / 1024) against torch: every kernel size / stride it serves, zero and reflection padding, transposed layers,
    ragged patches; the accumulate-into-bucket form (beta = 1) as well."""
    from iprgan import _lib, ops
    cin, cout, k, s, p, op, tr, pm, H, W, B = shape
    x = rnd(B, cin, H, W, seed=1)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.zeros(*wshape, requires_grad=True)
    if tr:
        y = F.conv_transpose2d(x, w, None, stride=s, padding=p, output_padding=op)
    elif pm:
        y = F.conv2d(F.pad(x, (p, p, p, p), mode='reflect'), w, None, stride=s)
    else:
        y = F.conv2d(x, w, None, stride=s, padding=p)
    g = rnd(*y.shape, seed=4)
    y.backward(g)
    try:
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        spec = ops.ConvSpec(cin, cout, k, s, p, op, tr, pad_mode=pm)
        d = spec.desc(B, H, W)
        xd, gd = to_nhwc(x).to(dev), to_nhwc(g).to(dev)
        dw, _ = ops.conv_bwd_weight(spec, d, xd, gd, wshape, False)
        close(dw, w.grad, what=f'fp32 halo wgrad cand {cand}')
        acc = torch.full(wshape, 0.5, device=dev)
        ops.conv_bwd_weight(spec, d, xd, gd, wshape, False, dw=acc, beta=1.0)
        close(acc.cpu(), w.grad + 0.5, what=f'fp32 halo wgrad beta=1 cand {cand}')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)


RGB_WGRAD_SHAPES = [  # cin, cout, transposed, H, W, B
    (3, 64, False, 16, 16, 3),        # stem 3 -> 64: tiles of 16 rows, halo rows above / below the image are zero padding
    (3, 128, False, 12, 32, 2),       # two 64-channel tiles of dy; H not a multiple of the tile's rows (ragged last tile)
    (64, 3, True, 8, 64, 2),          # head 64 -> 3 (ConvTranspose2d): T64 = x, T4 = dy
    (3, 64, False, 6, 128, 2),        # full-width rows of config 5 (two rows per tile)
]


@pytest.mark.gpu
@pytest.mark.parametrize('cand', [69, 70])
@pytest.mark.parametrize('shape', RGB_WGRAD_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_wgrad_rgb(dev, shape, cand):
    """Backward-weight of the RGB layers in the streaming form (wgrad_rgb_kernel; candidates 69 / 70) with the
    many-channel tensor stored as bf16 and the 3-channel side fp32: the kernel rounds the fp32 side to bf16 (math-mode
    operand rounding), so the reference is computed from the rounded image and the products are exact."""
    from iprgan import _lib, ops
    cin, cout, tr, H, W, B = shape
    k, s, p = 3, 1, 1
    x = rnd(B, cin, H, W, seed=1).bfloat16().float()
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.zeros(*wshape, requires_grad=True)
    y = F.conv_transpose2d(x, w, None, stride=s, padding=p) if tr else F.conv2d(x, w, None, stride=s, padding=p)
    g = rnd(*y.shape, seed=4).bfloat16().float()
    y.backward(g)
    try:
        _lib.set_math('bf16act')
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
        d = spec.desc(B, H, W)
        xd, gd = to_nhwc(x).to(dev), to_nhwc(g).to(dev)
        if tr:
            xd = xd.bfloat16()
        else:
            gd = gd.bfloat16()
        dw, _ = ops.conv_bwd_weight(spec, d, xd, gd, wshape, False)
        close(dw, w.grad, 2e-4, f'rgb wgrad cand {cand}')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


BF16_SHAPES = [  # cin, cout, k, s, p, transposed, H, B
    (64, 128, 3, 1, 1, False, 16, 4), (128, 64, 4, 2, 1, False, 16, 4), (256, 128, 4, 2, 1, True, 8, 4),
    (32, 96, 3, 1, 1, False, 9, 3), (64, 64, 3, 2, 1, False, 17, 2),
    # RGB stems / heads: the few-input-channel MFMA kernel (forward of 3->C, backward-data of C->3), ragged tiles
    (3, 64, 3, 1, 1, False, 19, 3), (3, 96, 4, 2, 1, False, 18, 2), (64, 3, 3, 1, 1, True, 13, 2), (3, 64, 7, 1, 3, False, 20, 2)]


@pytest.mark.parametrize('shape', BF16_SHAPES)
def test_conv_bf16_math_mode(dev, shape):
    """IPRGAN_MATH_BF16: operands rounded to bf16 (nearest-even) in LDS, fp32 accumulation.  Checked two ways:
    (1) against fp32 torch on inputs that are ALREADY bf16-representable - then the products are exact and only the
    summation order differs (2e-4 of max, like the fp32 kernels); (2) against fp32 torch on generic inputs within
    the bf16 rounding bound 2^-8 * sqrt(2) relative to max|x| max|w| sqrt(K)."""
    from iprgan import _lib, ops
    cin, cout, k, s, p, tr, H, B = shape
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
    d = spec.desc(B, H, H)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    conv = (lambda x, w: F.conv_transpose2d(x, w, None, stride=s, padding=p)) if tr else \
           (lambda x, w: F.conv2d(x, w, None, stride=s, padding=p))
    try:
        _lib.set_math('bf16')
        assert _lib.get_math() == 'bf16'
        for exact in (True, False):
            x, w = rnd(B, cin, H, H, seed=1), rnd(*wshape, seed=2, scale=0.05)
            if exact:
                x, w = x.bfloat16().float(), w.bfloat16().float()
            xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
            yr = conv(xr, wr)
            g = rnd(*yr.shape, seed=3)
            if exact:
                g = g.bfloat16().float()
            yr.backward(g)
            wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
            y = ops.conv_fwd(spec, d, to_nhwc(x).to(dev), wf, None)
            dx = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb)
            dw, _ = ops.conv_bwd_weight(spec, d, to_nhwc(x).to(dev), to_nhwc(g).to(dev), wshape, False)
            tol = 2e-4 if exact else 2e-2
            close(from_nhwc(y.cpu(), cout), yr, tol, f'bf16 fwd exact={exact}')
            close(from_nhwc(dx.cpu(), cin), xr.grad, tol, f'bf16 dgrad exact={exact}')
            close(dw, wr.grad, tol, f'bf16 wgrad exact={exact}')
    finally:
        _lib.set_math('fp32')


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['bf16', 'bf16act'])
@pytest.mark.parametrize('tile', [0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 17])
def test_bf16_gconv_tiles(dev, tile, mode):
    """Every bf16 tile of the forward / backward-data kernel, including the 128x64 and 128x128 per-wave tiles that only
    the bf16 modes build (6, 7), on bf16-representable inputs (products exact: only the summation order differs), with
    fp32 and with bf16 activations in HBM; ragged M (not a multiple of 256) and N = 256 so that 256-wide tiles apply.
    Tiles 8-15 and 17 are the LDS-DMA ring tiles (14, 15: the persistent form, 17: the half-tile ring) of conv_pipe.hip
    (bf16 operands in HBM only)."""
    from iprgan import _lib, ops
    if tile >= 8 and mode != 'bf16act':
        pytest.skip('the LDS-DMA ring tiles read bf16 operands from HBM')
    cin, cout, k, s, p, H, W, B = 64, 256, 3, 1, 1, 15, 13, 5
    x, w = rnd(B, cin, H, W, seed=1).bfloat16().float(), rnd(cout, cin, k, k, seed=2, scale=0.05).bfloat16().float()
    b = rnd(cout, seed=5, scale=0.3)
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    yr = F.conv2d(xr, wr, b, stride=s, padding=p)
    g = rnd(*yr.shape, seed=3).bfloat16().float()
    yr.backward(g)
    try:
        _lib.set_math(mode)
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        spec = ops.ConvSpec(cin, cout, k, s, p)
        d = spec.desc(B, H, W)
        assert bool(d.x_bf16) == (mode == 'bf16act')
        wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
        y = ops.conv_fwd(spec, d, to_nhwc(x).to(dev), wf, b.to(dev))
        dx = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb)
        # outputs stored as bf16 carry one more rounding (2^-9 relative)
        tol = 2e-4 if mode == 'bf16' else 6e-3
        close(from_nhwc(y.float().cpu(), cout), yr, tol, f'{mode} fwd tile {tile}')
        close(from_nhwc(dx.float().cpu(), cin), xr.grad, tol, f'{mode} dgrad tile {tile}')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


@pytest.mark.gpu
@pytest.mark.parametrize('tile', [8, 9, 11, 13, 14, 15])
def test_pipe_tiles_fp32_reflect_and_epilogue(dev, tile):
    """fp32 form of the LDS-DMA ring tiles: ReflectionPad2d folded into the DMA offsets (CycleGAN's residual convolutions,
    networks/resnet_generator.py:26-29), bias + LeakyReLU, epilogue column sums, the fused activation derivative with a
    residual input, and the paired pass - against torch, fp32 tolerance."""
    from iprgan import _lib, ops
    cin, cout, k, p, H, W, B = 64, 128, 3, 1, 13, 11, 4
    x = rnd(B, cin, H, W, seed=1)
    w, b = rnd(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5), rnd(cout, seed=3, scale=0.3)
    try:
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        _lib.call('iprgan_debug_force_splitk', 1)
        for pm in (1, 0):
            conv = (lambda t: F.conv2d(F.pad(t, (p, p, p, p), mode='reflect'), w, None)) if pm else \
                   (lambda t: F.conv2d(t, w, None, padding=p))
            xin = F.leaky_relu(x, 0.2).requires_grad_()
            acc = conv(xin)
            y_ref = F.leaky_relu(acc + b.view(1, -1, 1, 1), 0.1)
            spec = ops.ConvSpec(cin, cout, k, 1, p, 0, False, pad_mode=pm, act=2, slope=0.1)
            d = spec.desc(B, H, W)
            wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
            xd = to_nhwc(xin.detach()).to(dev)
            y, (part, rows) = ops.conv_fwd(spec, d, xd, wf, b.to(dev), stats=True)
            close(from_nhwc(y.cpu(), cout), y_ref.detach(), what=f'fp32 pipe fwd tile {tile} pad_mode {pm}')
            Cs = ops.c4(cout)
            pr = part[:rows * 2 * Cs].view(rows, 2, Cs).double().sum(0).cpu()
            a64 = acc.detach().double()
            assert float((pr[0, :cout] - a64.sum((0, 2, 3))).abs().max()) <= 2e-4 * float(acc.abs().max()) * acc[:, 0].numel() ** 0.5
            s0, s1 = torch.tensor([1.5], device=dev), torch.tensor([0.75], device=dev)
            yp = ops.conv_fwd(spec, d, xd, wf, b.to(dev), pair=(s0, s1))
            accp = acc.detach().clone()
            accp[:B // 2] /= 1.5
            accp[B // 2:] /= 0.75
            close(from_nhwc(yp.cpu(), cout), F.leaky_relu(accp + b.view(1, -1, 1, 1), 0.1), what=f'fp32 pipe pair tile {tile}')
            if not pm:          # backward-data of the zero-padded layer through the same tiles (reflect goes through the fold)
                g = rnd(*acc.shape, seed=4)
                res = rnd(*x.shape, seed=5)
                acc.backward(g)
                dx_ref = xin.grad * torch.where(xin > 0, 1.0, 0.2) + res
                dspec = ops.ConvSpec(cin, cout, k, 1, p, 0, False)
                dx, (part, rows) = ops.conv_bwd_data(dspec, dspec.desc(B, H, W), to_nhwc(g).to(dev), wb, xd, 2, 0.2,
                                                     colsums=True, residual=to_nhwc(res).to(dev))
                close(from_nhwc(dx.cpu(), cin), dx_ref, what=f'fp32 pipe dgrad tile {tile}')
                cs = ops.colsum_partials(part, rows, ops.c4(cin), cin).cpu().double()
                assert float((cs - dx_ref.double().sum((0, 2, 3))).abs().max()) <= 2e-4 * float(dx_ref.abs().max()) * dx_ref[:, 0].numel() ** 0.5
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.call('iprgan_debug_force_splitk', -1)


PIPE_SHAPES = [  # cin, cout, k, s, p, outpad, transposed, H, W, B
    (64, 128, 3, 1, 1, 0, False, 15, 13, 5),      # k3: 9 taps x one 64-deep step; ragged M
    (64, 64, 4, 2, 1, 0, False, 18, 14, 3),       # D.conv1 form: forward one phase of 16 steps, backward-data four phases of 4
    (128, 64, 4, 2, 1, 0, True, 8, 8, 6),         # ConvT k4s2 (generator): forward = four sub-pixel phases
    (64, 64, 3, 2, 1, 1, True, 6, 5, 2),          # ConvT k3s2 with output_padding: phases of 1, 2, 2, 4 taps (one-step rings)
    (256, 256, 4, 2, 1, 0, False, 8, 8, 4),       # two channel steps per tap, N = 256 (the 256x256 tile applies)
    (64, 256, 3, 2, 1, 1, True, 6, 5, 2),         # N = 256 with phases of 1, 2, 2, 4 K tiles (prologue / tail of the half-tile ring)
    (128, 256, 3, 1, 1, 0, False, 9, 7, 6),       # N = 256, 18 K tiles, ragged M (two tile rows)
    # the four-phases-per-block tile (16): whole grid rows x 256 positions; halo rows above / below the image are padding
    (64, 64, 4, 2, 1, 0, False, 128, 128, 1),     # backward-data: 64 x 64 grid, one channel chunk (single halo buffer)
    (128, 64, 4, 2, 1, 0, True, 32, 32, 2),       # ConvT forward: 32 x 32 grid, two chunks (double-buffered halo), Cout 64
    (64, 128, 4, 2, 1, 0, False, 32, 32, 2),      # backward-data with Cout = 128 reduced in two chunks, 16 x 16 grid
]


@pytest.mark.gpu
@pytest.mark.parametrize('tile', [8, 9, 10, 11, 12, 13, 14, 15, 16, 17])
@pytest.mark.parametrize('shape', PIPE_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_pipe_tiles(dev, shape, tile):
    """The LDS-DMA ring tiles (conv_pipe.hip) on bf16-representable operands (products exact, fp32 accumulation: only the
    summation order differs from torch) through every geometry form they serve: strided forward, sub-pixel phases of the
    transposed / backward-data forms incl. phases shorter than the ring, zero padding by out-of-range DMA offsets, ragged
    M; with bias + LeakyReLU, the fused activation derivative, epilogue column sums and the paired pass.  A tile that does
    not apply to a geometry (N below its width) falls back to the register-staged kernel, which keeps the check valid."""
    from iprgan import _lib, ops
    cin, cout, k, s, p, op, tr, H, W, B = shape
    x = rnd(B, cin, H, W, seed=1).bfloat16().float()
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = rnd(*wshape, seed=2, scale=(cin * k * k) ** -0.5).bfloat16().float()
    b = rnd(cout, seed=3, scale=0.3)
    conv = (lambda t: F.conv_transpose2d(t, w, None, stride=s, padding=p, output_padding=op)) if tr else \
           (lambda t: F.conv2d(t, w, None, stride=s, padding=p))
    xin = F.leaky_relu(x, 0.25).bfloat16().float().requires_grad_()       # slope 1/4: stays bf16-representable
    acc = conv(xin)
    y_ref = F.leaky_relu(acc + b.view(1, -1, 1, 1), 0.1)
    g = rnd(*acc.shape, seed=4).bfloat16().float()
    acc.backward(g)
    dx_ref = xin.grad * torch.where(xin > 0, 1.0, 0.25)
    try:
        _lib.set_math('bf16act')
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        spec = ops.ConvSpec(cin, cout, k, s, p, op, tr, act=2, slope=0.1)
        d = spec.desc(B, H, W)
        assert d.x_bf16 and d.y_bf16
        wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
        xd = to_nhwc(xin.detach()).to(dev).bfloat16()
        y, (part, rows) = ops.conv_fwd(spec, d, xd, wf, b.to(dev), stats=True)
        close(from_nhwc(y.float().cpu(), cout), y_ref.detach(), 6e-3, f'pipe fwd tile {tile}')
        Cs = ops.c4(cout)
        pr = part[:rows * 2 * Cs].view(rows, 2, Cs).double().sum(0).cpu()
        a64 = acc.detach().double()
        assert float((pr[0, :cout] - a64.sum((0, 2, 3))).abs().max()) <= 2e-4 * float(acc.abs().max()) * acc[:, 0].numel() ** 0.5
        assert float((pr[1, :cout] - (a64 ** 2).sum((0, 2, 3))).abs().max()) <= 2e-4 * float((a64 ** 2).sum((0, 2, 3)).max())
        dspec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
        dd = dspec.desc(B, H, W)
        gd = to_nhwc(g).to(dev).bfloat16()
        dx, (part, rows) = ops.conv_bwd_data(dspec, dd, gd, wb, xd, 2, 0.25, colsums=True)
        close(from_nhwc(dx.float().cpu(), cin), dx_ref, 6e-3, f'pipe dgrad tile {tile}')
        cs = ops.colsum_partials(part, rows, ops.c4(cin), cin).cpu().double()
        # the sums are taken over the fp32 values before they are rounded to bf16 for the store
        assert float((cs - dx_ref.double().sum((0, 2, 3))).abs().max()) <= 2e-4 * float(dx_ref.abs().max()) * dx_ref[:, 0].numel() ** 0.5
        if B % 2 == 0:          # paired pass: rows of each half-batch divided by its own sigma before the bias
            s0, s1 = torch.tensor([1.5], device=dev), torch.tensor([0.75], device=dev)
            yp = ops.conv_fwd(spec, d, xd, wf, b.to(dev), pair=(s0, s1))
            accp = acc.detach().clone()
            accp[:B // 2] /= 1.5
            accp[B // 2:] /= 0.75
            close(from_nhwc(yp.float().cpu(), cout), F.leaky_relu(accp + b.view(1, -1, 1, 1), 0.1), 6e-3, f'pipe pair tile {tile}')
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


NORTH_STAR = [
    # SURVEY 8d "conv microbench (north-star)": the 3x3 convs Resnet9Blocks runs on a 64x3x256x256 batch
    # cin, cout, stride, pad_mode, H
    (64, 128, 2, 0, 256),
    (128, 256, 2, 0, 128),
    (256, 256, 1, 1, 64),          # the headline shape: reflect-padded 256->256 on (64,256,64,64)
]


@pytest.mark.parametrize('cfg', NORTH_STAR, ids=lambda c: '-'.join(map(str, c)))
def test_north_star_conv_shapes_full_size(dev, cfg):
    """Forward, backward-data and backward-weight at the FULL north-star sizes (batch 64) against torch CPU.  Forward
    and backward-data are per-image, so three images of the batch are checked against the CPU (first, middle, last:
    tile edges at both ends of the M range); backward-weight reduces over all 64 images and is checked in full."""
    from iprgan import ops
    cin, cout, s, pm, H = cfg
    B, k, p = 64, 3, 1
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, cin, H, H, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) * (cin * k * k) ** -0.5
    spec = ops.ConvSpec(cin, cout, k, s, p, pad_mode=pm)
    d = spec.desc(B, H, H)
    OH, OW = spec.out_hw(H, H)
    gy = torch.randn(B, cout, OH, OW, generator=g)

    def ref_fwd(xx):
        return F.conv2d(F.pad(xx, (p, p, p, p), mode='reflect'), w, None, stride=s) if pm else F.conv2d(xx, w, None, stride=s, padding=p)

    xd, gd = to_nhwc(x).to(dev), to_nhwc(gy).to(dev)
    wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
    y = ops.conv_fwd(spec, d, xd, wf, None)
    dx = ops.conv_bwd_data(spec, d, gd, wb)
    dw, _ = ops.conv_bwd_weight(spec, d, xd, gd, w.shape, False)
    sel = [0, B // 2, B - 1]
    xs = x[sel].clone().requires_grad_()
    ys = ref_fwd(xs)
    ys.backward(gy[sel])
    close(from_nhwc(y[sel].cpu(), cout), ys, what='fwd')
    close(from_nhwc(dx[sel].cpu(), cin), xs.grad, what='dgrad')
    # full backward-weight on the CPU: 2*B*OH*OW*cout*cin*9 FLOP (154-309 GFLOP), seconds on the host cores
    wr = w.clone().requires_grad_()
    xin = F.pad(x, (p, p, p, p), mode='reflect') if pm else x
    dwr = torch.nn.grad.conv2d_weight(xin, w.shape, gy, stride=s, padding=0 if pm else p)
    close(dw, dwr, what='wgrad')


STAT_SHAPES = [  # cin, cout, k, s, p, outpad, transposed, H, W, B
    (64, 64, 3, 1, 1, 0, False, 16, 16, 4),       # 256 rows per sample
    (128, 64, 4, 2, 1, 0, True, 8, 8, 6),         # ConvT k4s2: four sub-pixel phases of 64 rows per sample
    (32, 96, 3, 2, 1, 0, False, 15, 13, 3),       # ragged rows, Cout not a tile multiple
    (64, 32, 3, 2, 1, 1, True, 6, 5, 2),          # ConvT with output_padding: phases of different sizes
]


@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize('shape', STAT_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_conv_epilogue_column_sums(dev, shape, tile):
    """Column sums emitted by the conv epilogues, under every tile variant: forward (sum acc, sum acc^2 of the pre-bias
    accumulator per tile row -> reduced here over all rows) against torch; backward-data (column sums of dx after the
    fused activation derivative) against torch.  Also: outputs are unchanged by the extra epilogue work."""
    from iprgan import _lib, ops
    cin, cout, k, s, p, op, tr, H, W, B = shape
    x = rnd(B, cin, H, W, seed=1)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w, b = rnd(*wshape, seed=2, scale=(cin * k * k) ** -0.5), rnd(cout, seed=3, scale=0.3)
    conv = (lambda t: F.conv_transpose2d(t, w, None, stride=s, padding=p, output_padding=op)) if tr else \
           (lambda t: F.conv2d(t, w, None, stride=s, padding=p))
    acc = conv(x)
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr)
    d = spec.desc(B, H, W)
    try:
        _lib.call('iprgan_debug_force_tiles', tile, -1)
        _lib.call('iprgan_debug_force_splitk', 1)       # bit-equality below: the plain pass must not take the split-K path
        wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
        xd = to_nhwc(x).to(dev)
        y0 = ops.conv_fwd(spec, d, xd, wf, b.to(dev))
        y, (part, rows) = ops.conv_fwd(spec, d, xd, wf, b.to(dev), stats=True)
        if tile >= 0:
            assert torch.equal(y, y0)
        else:   # the tuner times the two epilogues apart and may pick tiles whose MFMA shapes sum a K step in a different order
            assert float((y - y0).abs().max()) <= 2e-6 * float(y0.abs().max())
        Cs = ops.c4(cout)
        pr = part[:rows * 2 * Cs].view(rows, 2, Cs).double().sum(0).cpu()
        ref1, ref2 = acc.double().sum((0, 2, 3)), (acc.double() ** 2).sum((0, 2, 3))
        assert float((pr[0, :cout] - ref1).abs().max()) <= 2e-4 * float(acc.abs().max()) * acc[:, 0].numel() ** 0.5
        assert float((pr[1, :cout] - ref2).abs().max()) <= 2e-4 * float(ref2.abs().max())
        # backward-data: dx * lrelu'(x_in) summed over all pixels = bias gradient of a producer conv with LeakyReLU
        g = rnd(*acc.shape, seed=4)
        xin = F.leaky_relu(x, 0.1).requires_grad_()
        conv(xin).backward(g)
        want = (xin.grad * torch.where(xin > 0, 1.0, 0.1))
        xin_d = to_nhwc(xin.detach()).to(dev)
        dx0 = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb, xin_d, 2, 0.1)
        dx, (part, rows) = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb, xin_d, 2, 0.1, colsums=True)
        if tile >= 0:
            assert torch.equal(dx, dx0)
        else:
            assert float((dx - dx0).abs().max()) <= 2e-6 * float(dx0.abs().max())
        cs = ops.colsum_partials(part, rows, ops.c4(cin), cin).cpu().double()
        refc = want.double().sum((0, 2, 3))
        assert float((cs - refc).abs().max()) <= 2e-4 * float(want.abs().max()) * want[:, 0].numel() ** 0.5
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.call('iprgan_debug_force_splitk', -1)


@pytest.mark.parametrize('ratio', [3.0, 20.0, 60.0])
def test_fused_norm_statistics_with_large_mean_channels(dev, ratio):
    """the statistics a BatchNorm takes from the producing conv's epilogue are sums about ZERO of the pre-bias
    accumulator (s2 / M - (s1 / M)^2), not shifted sums: fp32 cancellation grows with (mean / std)^2 of a channel.  A
    convolution whose output channels sit ``ratio`` standard deviations away from zero (positive weights on a positive
    input), normalised by the fused path, against torch.batch_norm in float64: the normalised output must stay within
    1e-6 * (1 + ratio^2) of the output scale (measured: 2.5e-6 / 9.0e-5 / 7.6e-4 for ratio 3 / 20 / 60, i.e. a quarter of
    the bound; the generator / discriminator layers of the three GANs sit below ratio 3) and invstd within the same
    relative bound."""
    from iprgan import ops
    cin, cout, k, H, W, B = 64, 64, 3, 34, 34, 4          # unpadded: a zero border would put a floor under the channel's std
    x0 = rnd(B, cin, H, W, seed=1, scale=0.1)
    w = 0.01 + rnd(cout, cin, k, k, seed=2, scale=0.0005)        # nearly equal row sums: one input offset moves every channel alike
    sw = float(w.double().sum((1, 2, 3)).median())
    c = 0.0
    for _ in range(4):                                            # input offset that puts the median channel at `ratio` sigma
        acc = F.conv2d((x0 + c).double(), w.double(), None)
        m, sd = acc.mean((0, 2, 3)), acc.std((0, 2, 3))
        c += float(((ratio * sd - m) / sw).median())
    x = x0 + c
    acc = F.conv2d(x.double(), w.double(), None)
    got_ratio = (acc.mean((0, 2, 3)).abs() / acc.std((0, 2, 3)))
    gamma, beta = rnd(cout, seed=3, scale=0.3) + 1.0, rnd(cout, seed=4, scale=0.2)
    want = F.batch_norm(acc, None, None, gamma.double(), beta.double(), training=True, eps=1e-5)
    spec = ops.ConvSpec(cin, cout, k, 1, 0)
    d = spec.desc(B, H, W)
    wf, _ = ops.conv_prep(spec, d, w.to(dev), None, True, False)
    y, stats = ops.conv_fwd(spec, d, to_nhwc(x).to(dev), wf, None, stats=True)
    out, mean_d, invstd_d = ops.bn_fwd(y, gamma.to(dev), beta.to(dev), None, None, 1e-5, 0.0, True, 0, 0.0, conv_stats=stats)
    bound = 1e-6 * (1.0 + float(got_ratio.max()) ** 2)
    err = float((from_nhwc(out.cpu(), cout).double() - want).abs().max())
    print(f'large-mean statistics: mean/std {float(got_ratio.min()):.1f}..{float(got_ratio.max()):.1f}: output error {err / float(want.abs().max()):.2e} of the output scale (bound {bound:.2e})')
    assert err <= bound * float(want.abs().max()), (err, bound, float(got_ratio.min()), float(got_ratio.max()))
    var = acc.var((0, 2, 3), unbiased=False)
    assert float(((invstd_d.cpu().double() - (var + 1e-5).rsqrt()) / (var + 1e-5).rsqrt()).abs().max()) <= bound


SPLITK_SHAPES = [   # cin, cout, k, stride, pad, transposed, H, W, B, pad_mode: few output tiles, long reductions
    (512, 512, 3, 1, 1, False, 6, 6, 16, 0),      # VGG block 5 of the SRGAN content loss (networks/vgg.py) at a small batch
    (256, 128, 3, 2, 1, False, 12, 10, 8, 0),     # strided: still one phase in the forward pass; backward is 4 phases
    (128, 192, 3, 1, 1, False, 9, 7, 6, 1),       # reflect padding forward (backward goes through the fold, not split)
    (256, 64, 4, 2, 1, True, 5, 6, 8, 0),         # transposed: the backward-data pass is the single-phase one
]


@pytest.mark.gpu
@pytest.mark.parametrize('splits', [1, 2, 3, 4])
@pytest.mark.parametrize('shape', SPLITK_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_conv_splitk(dev, shape, splits):
    """Split-K path of regular convolutions with few output tiles (forced split counts): forward with bias + LeakyReLU
    and backward-data with the fused activation derivative and a residual input against torch; every split count must
    agree with the unsplit kernel to fp32 rounding."""
    from iprgan import _lib, ops
    cin, cout, k, s, p, tr, H, W, B, pm = shape
    x = rnd(B, cin, H, W, seed=1)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w, b = rnd(*wshape, seed=2, scale=(cin * k * k) ** -0.5), rnd(cout, seed=3, scale=0.3)
    if tr:
        conv = lambda t: F.conv_transpose2d(t, w, b, stride=s, padding=p)
    elif pm:
        conv = lambda t: F.conv2d(F.pad(t, (p, p, p, p), mode='reflect'), w, b, stride=s)
    else:
        conv = lambda t: F.conv2d(t, w, b, stride=s, padding=p)
    xin = F.leaky_relu(x, 0.2).requires_grad_()
    y_ref = F.leaky_relu(conv(xin), 0.1)
    g = rnd(*y_ref.shape, seed=4)
    res = rnd(*x.shape, seed=5)
    conv(xin).backward(g)
    dx_ref = xin.grad * torch.where(xin > 0, 1.0, 0.2) + res
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=pm, act=2, slope=0.1)
    d = spec.desc(B, H, W)
    import ctypes as C
    assert _lib.query('iprgan_conv_fwd_ws_floats', C.byref(d)) > 0 or _lib.query('iprgan_conv_bwd_data_ws_floats', C.byref(d)) > 0
    try:
        _lib.call('iprgan_debug_force_splitk', splits)
        wf, wb = ops.conv_prep(spec, d, w.to(dev), None, True, True)
        xd = to_nhwc(xin.detach()).to(dev)
        y = ops.conv_fwd(spec, d, xd, wf, b.to(dev))
        close(from_nhwc(y.cpu(), cout), y_ref.detach(), what='split-K forward')
        dspec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=pm)
        dx = ops.conv_bwd_data(dspec, dspec.desc(B, H, W), to_nhwc(g).to(dev), wb, xd, 2, 0.2, residual=to_nhwc(res).to(dev))
        close(from_nhwc(dx.cpu(), cin), dx_ref, what='split-K backward-data')
    finally:
        _lib.call('iprgan_debug_force_splitk', -1)


@pytest.mark.parametrize('shape,denorm', [((2, 3, 176, 193), False), ((1, 3, 256, 256), True), ((2, 1, 161, 200), False)])
def test_ms_ssim_loss_vs_oracle(dev, shape, denorm):
    """tools.ms_ssim (tools/loss.py:78-80) on the HIP kernels against the restated pytorch-msssim algorithm: value and
    gradient w.r.t. x, odd sizes (zero-padded pooling) included."""
    from iprgan import tools
    from oracle import ssim as ossim
    g = torch.Generator().manual_seed(5)
    x = torch.rand(*shape, generator=g)
    y = (x + 0.3 * (torch.rand(*shape, generator=g) - 0.5)).clamp(0, 1)
    if denorm:
        x, y = x * 2 - 1, y * 2 - 1
    xa = x.clone().requires_grad_()
    xb = x.clone().to(dev).requires_grad_()
    la = ossim.ms_ssim_loss(normalized=denorm)(xa, y)
    lb = tools.ms_ssim(normalized=denorm)(xb, y.to(dev))
    np.testing.assert_allclose(float(lb.detach()), float(la.detach()), rtol=2e-4, atol=2e-6)
    (3.0 * la).backward(); (3.0 * lb).backward()
    ga, gb = xa.grad.double(), xb.grad.cpu().double()
    assert float((ga - gb).abs().max()) <= 2e-3 * float(ga.abs().max()), float((ga - gb).abs().max()) / float(ga.abs().max())
    # MS-SSIM(x, x) = 1 -> loss 0
    assert abs(float(tools.ms_ssim(normalized=denorm)(xb.detach(), xb.detach()))) < 2e-6


@pytest.mark.gpu
def test_integration_md_ctypes_stub(dev):
    """The ctypes stub INTEGRATION.md shows a reference maintainer (SN conv 3x3 + LeakyReLU through the C ABI only) is
    executed as written and compared with torch's spectral_norm + conv2d + leaky_relu on the CPU."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    block = next(b for b in re.findall(r"```python\n(.*?)```", text, re.S) if 'def sn_conv3x3_lrelu' in b)
    ns = {}
    exec(block.replace('<repo>', root), ns)
    conv = torch.nn.utils.spectral_norm(torch.nn.Conv2d(8, 16, 3, 1, 1))
    x = rnd(2, 8, 9, 7, seed=3)
    with torch.no_grad():
        w0, u0, v0 = conv.weight_orig.clone(), conv.weight_u.clone(), conv.weight_v.clone()
        conv.train()
        ref = F.leaky_relu(conv(x), 0.1)                      # one power iteration, then W / sigma
    y = ns['sn_conv3x3_lrelu'](x.to(dev), w0.to(dev), u0.to(dev), v0.to(dev), conv.bias.detach().to(dev))
    torch.cuda.synchronize()
    close(from_nhwc(y.cpu(), 16), ref, what='INTEGRATION.md stub')


@pytest.mark.parametrize('n_img,pool_n,shape', [(8, 50, (3, 32, 32)), (1, 4, (3, 7, 5)), (70, 80, (1, 3, 3))])
def test_pool_swap_and_write_ints(dev, n_img, pool_n, shape):
    """iprgan_write_ints + iprgan_pool_swap (ImagePool's swap branch, models/util.py:27-34, with the draws in device memory):
    the same result as the reference's three indexed copies, for odd sizes (scalar tail) and more than 64 entries (two
    launches of the by-value writer)."""
    from iprgan import ops
    g = torch.Generator().manual_seed(11 + n_img)
    images = torch.randn(n_img, *shape, generator=g)
    pool = torch.randn(pool_n, *shape, generator=g)
    prob = torch.rand(n_img, generator=g) > 0.5
    index = torch.randperm(pool_n, generator=g)[:n_img]
    want_img, want_pool = images.clone(), pool.clone()
    taken = want_pool[index[prob]].clone()
    want_pool[index[prob]] = want_img[prob]
    want_img[prob] = taken
    img_d, pool_d = images.to(dev), pool.to(dev)
    tbl = torch.zeros(2, n_img + 3, dtype=torch.int32, device=dev)
    ops.write_ints(tbl[0], index.tolist())
    ops.write_ints(tbl[1], [int(p) for p in prob.tolist()])
    assert tbl[0, :n_img].cpu().tolist() == index.tolist() and tbl[1, :n_img].cpu().tolist() == [int(p) for p in prob.tolist()]
    ops.pool_swap(img_d, pool_d, tbl[0], tbl[1])
    assert torch.equal(img_d.cpu(), want_img) and torch.equal(pool_d.cpu(), want_pool)


@pytest.mark.parametrize('cand', [-1, 0, 22, 45, 61, 72, 74, 76])
def test_deferred_wgrad_reduce_is_bit_identical(dev, cand):
    """iprgan_conv_bwd_weight_deferred + ONE iprgan_wgrad_reduce_multi over several layers against the per-layer form
    (tiles + reduce in one call): the same bits, with beta = 0 and accumulating into a pre-filled gradient (beta = 1), for the
    split-M GEMM forms, the halo forms of every storage kind and the autotuner's own choice (-1).  Every block of the multi
    kernel performs the additions of its single-layer counterpart in the same order (csrc/conv_igemm.hip: wgrad_reduce_block)."""
    from iprgan import _lib, ops
    layers = [(64, 64, 4, 2, 1, False, 32, 6), (64, 128, 3, 1, 1, False, 16, 5), (128, 64, 4, 2, 1, True, 8, 4),
              (96, 160, 3, 1, 1, False, 12, 3), (3, 64, 3, 1, 1, False, 16, 4)]
    try:
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        for beta in (0.0, 1.0):
            pending, want, got = [], [], []
            for li, (cin, cout, k, s, p, tr, H, B) in enumerate(layers):
                spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
                d = spec.desc(B, H, H)
                OH, OW = spec.out_hw(H, H)
                x = ops.to_kind(to_nhwc(rnd(B, cin, H, H, seed=10 + li)).to(dev), d.x_bf16)
                dy = ops.to_kind(to_nhwc(rnd(B, cout, OH, OW, seed=20 + li)).to(dev), d.y_bf16)
                wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
                base = rnd(*wshape, seed=30 + li).to(dev)
                a, b = base.clone(), base.clone()
                ops.conv_bwd_weight(spec, d, x, dy, wshape, False, dw=a, beta=beta)
                ops.conv_bwd_weight(spec, d, x, dy, wshape, False, dw=b, beta=beta, defer=pending)
                want.append(a); got.append(b)
            assert len(pending) >= 3, 'these layers run slab kernels: their reduces must have been deferred'
            ops.wgrad_reduce_flush(pending)
            assert not pending
            for li, (a, b) in enumerate(zip(want, got)):
                assert torch.equal(a, b), f'layer {li}, beta {beta}, cand {cand}: deferred reduce differs (max {float((a - b).abs().max()):.3e})'
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)


@pytest.mark.parametrize('cfg', [(64, 64, 3, 1, 16, 16, 3), (64, 128, 3, 1, 9, 11, 2), (128, 64, 5, 2, 12, 10, 2), (32, 64, 7, 3, 16, 9, 2),
                                 (256, 256, 3, 1, 64, 64, 2)], ids=lambda c: '-'.join(map(str, c)))
def test_reflect_pad_dgrad_without_the_padded_grid(dev, cfg):
    """conv_bwd_data_reflect_direct (csrc/conv_igemm.hip): the border strips of the padded gradient as four phases of one small
    launch, the image as an ordinary zero-padded backward-data pass with the fused LeakyReLU derivative and the skip-connection
    residual in its epilogue, reflect_ring_fix_kernel adding the mirrored strips to the ring pixels - against torch's gradient
    of ReflectionPad2d(p) + Conv2d(k = 2p + 1) in float64, for p = 1, 2, 3, ragged maps and the CycleGAN layer at full size."""
    from iprgan import _lib, ops
    cin, cout, k, p, H, W, B = cfg
    x = rnd(B, cin, H, W, seed=1)
    w = rnd(cout, cin, k, k, seed=2, scale=(cin * k * k) ** -0.5)
    prev = rnd(B, cin, H, W, seed=5)                     # the producer's output: its LeakyReLU derivative is fused into the epilogue
    res = rnd(B, cin, H, W, seed=6)
    xr = x.double().requires_grad_()
    yr = F.conv2d(F.pad(xr, (p, p, p, p), mode='reflect'), w.double(), None)
    g = rnd(*yr.shape, seed=4)
    yr.backward(g.double())
    want = xr.grad * torch.where(prev.double() > 0, 1.0, 0.2) + res.double()
    spec = ops.ConvSpec(cin, cout, k, 1, p, pad_mode=1)
    d = spec.desc(B, H, W)
    _, wb = ops.conv_prep(spec, d, w.to(dev), None, False, True)
    dx = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb, to_nhwc(prev).to(dev), _lib.ACT_LRELU, 0.2, residual=to_nhwc(res).to(dev))
    close(from_nhwc(ops.f32(dx).cpu(), cin), want, what='reflect dgrad, direct form')
    dx0 = ops.conv_bwd_data(spec, d, to_nhwc(g).to(dev), wb)
    close(from_nhwc(ops.f32(dx0).cpu(), cin), xr.grad, what='reflect dgrad, direct form, plain')


@pytest.mark.parametrize('mode', ['fp32', 'fp32x3', 'bf16act'])
@pytest.mark.parametrize('B,K,C,HW,act', [(128, 128, 512, 64, 1), (5, 32, 64, 3, 0), (130, 96, 128, 4, 2), (32, 64, 64, 1, 1)])
def test_fc_nhwc_direct_kernels(B, K, C, HW, act, mode, dev):
    """csrc/fc.hip: Linear(K -> C*HW) + activation feeding an NHWC map, forward and backward-weight in one launch each, on
    the parameters in PyTorch layout (networks/conv_generator.py:26-30: `self.fc(z).view(-1, C, mg, mg)`), against
    torch.nn.functional.linear on the CPU.  Every storage kind of the output (fp32 / bf16 / three planes), ragged batches
    (5 rows; 130 = two chunks), accumulation into an existing gradient (beta = 1).  Exact fp32 MFMA: 2e-4 of the tensor's
    scale as for the convolutions; a bf16 OUTPUT carries its own rounding (2^-8)."""
    from iprgan import _lib, ops
    _lib.set_math(mode)
    try:
        assert ops.fc_nhwc_ok(B, K, C, HW)
        N = C * HW
        x, w, b = rnd(B, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
        slope = 0.2
        z = F.linear(x, w, b)                                    # [B, N], PyTorch column order c*HW + hw
        yr = {0: z, 1: torch.relu(z), 2: F.leaky_relu(z, slope)}[act]
        nhwc = lambda t: t.view(B, C, HW).permute(0, 2, 1).reshape(B, N)          # noqa: E731  -> column order hw*C + c
        y = ops.fc_nhwc_fwd(x.to(dev), w.to(dev), b.to(dev), C, HW, act, slope)
        kind = ops.is16(y)
        assert kind == {'fp32': 0, 'fp32x3': 2, 'bf16act': 1}[mode] and tuple(y.shape) == (B, N)
        tol = 2e-4 if kind != 1 else 6e-3
        close(ops.f32(y), nhwc(yr), tol, 'fc forward')
        # backward: dy in the output's kind; the activation derivative is taken from the STORED y, as the engine does
        dy = rnd(B, N, seed=4)
        dyd = ops.to_kind(nhwc(dy).contiguous().to(dev), kind)
        ys = ops.f32(y).cpu().view(B, HW, C).permute(0, 2, 1).reshape(B, N)       # stored y back in PyTorch column order
        dyq = ops.f32(dyd).cpu().view(B, HW, C).permute(0, 2, 1).reshape(B, N)
        dz = dyq * {0: torch.ones_like(ys), 1: (ys > 0).float(), 2: torch.where(ys > 0, 1.0, slope)}[act]
        dw_ref, db_ref = dz.t() @ x, dz.sum(0)
        dw, db = ops.fc_nhwc_bwd(x.to(dev), y, dyd, (N, K), C, HW, act, slope)
        close(dw, dw_ref, 2e-4, 'fc dW')
        close(db, db_ref, 2e-4, 'fc db')
        # accumulate into existing gradients (the bucket views of the executor's gradient sink)
        g0, b0 = rnd(N, K, seed=5), rnd(N, seed=6)
        dw2, db2 = g0.clone().to(dev), b0.clone().to(dev)
        ops.fc_nhwc_bwd(x.to(dev), y, dyd, (N, K), C, HW, act, slope, dw=dw2, db=db2, beta=1.0)
        close(dw2, g0 + dw_ref, 2e-4, 'fc dW accumulate')
        close(db2, b0 + db_ref, 2e-4, 'fc db accumulate')
        # bit-for-bit repeatable
        dw3, db3 = ops.fc_nhwc_bwd(x.to(dev), y, dyd, (N, K), C, HW, act, slope)
        assert torch.equal(dw3, dw) and torch.equal(db3, db)
        assert torch.equal(ops.f32(ops.fc_nhwc_fwd(x.to(dev), w.to(dev), b.to(dev), C, HW, act, slope)), ops.f32(y))
    finally:
        _lib.set_math('fp32')


def test_fc_nhwc_refuses_unsupported_shapes(dev):
    from iprgan import _lib, ops
    assert not ops.fc_nhwc_ok(8, 2048, 128, 1)          # the VAE encoder heads: K too deep for the direct kernel's LDS tile
    assert not ops.fc_nhwc_ok(8, 128, 96, 4)            # C not a multiple of 64
    with pytest.raises(RuntimeError, match='unsupported shape'):
        _lib.call('iprgan_fc_nhwc_fwd', None, None, None, None, 8, 2048, 128, 1, 0, 0.0, 0, 0, None)


@pytest.mark.parametrize('norm', ['batch', 'instance', 'batch_prelu'])
def test_norm_backward_with_an_fp32_gradient_is_bit_identical_to_the_three_plane_one(norm, dev):
    """IPRGAN_ST_X3_XDF32 (round 6): on the conv -> norm edges of a backward pass the gradient reaches the norm layer as fp32
    (the backward-data pass writes 4 instead of 6 bytes per element) and the norm backward emits the three-plane dx.  A
    three-plane tensor holds its fp32 values exactly, so the result must equal, bit for bit, the one from the same gradient
    handed over as three planes (IPRGAN_ST_X3_XF32)."""
    from iprgan import _lib, ops
    _lib.set_math('fp32x3')
    try:
        B, H, W, C = 6, 12, 10, 96
        x = (rnd(B, H, W, C, seed=1) * 1.7 + 0.2).to(dev)
        dy = rnd(B, H, W, C, seed=2).to(dev)
        gamma, beta = (rnd(C, seed=3, scale=0.5) + 1.0).to(dev), rnd(C, seed=4, scale=0.2).to(dev)
        dyp = ops.to_kind(dy, ops.ST_X3)
        assert torch.equal(ops.f32(dyp), dy)
        if norm == 'batch':
            y, mean, invstd = ops.bn_fwd(x, gamma, beta, None, None, 1e-5, 0.0, True, _lib.ACT_RELU)
            run = lambda g: ops.bn_bwd(x, y, g, gamma, mean, invstd, _lib.ACT_RELU, beta=beta,            # noqa: E731
                                       dbias=torch.zeros(C, device=dev))
        elif norm == 'instance':
            y, mean, invstd = ops.instnorm_fwd(x, gamma, beta, 1e-5, _lib.ACT_LRELU, 0.2)
            run = lambda g: ops.instnorm_bwd(x, y, g, gamma, mean, invstd, _lib.ACT_LRELU, 0.2, beta=beta)     # noqa: E731
        else:
            alpha = torch.tensor([0.25], device=dev)
            y, mean, invstd = ops.bn_prelu_fwd(x, gamma, beta, None, None, 1e-5, 0.0, True, alpha)
            run = lambda g: ops.bn_prelu_bwd(x, g, gamma, beta, mean, invstd, alpha)                       # noqa: E731
        assert ops.is16(y) == ops.ST_X3
        a, b = run(dyp), run(dy)
        assert ops.is16(a[0]) == ops.ST_X3 and ops.is16(b[0]) == ops.ST_X3, 'dx leaves as three planes either way'
        for ta, tb, name in zip(a, b, ('dx', 'dgamma', 'dbeta', 'dslope')):
            assert torch.equal(ops.f32(ta), ops.f32(tb)), name
    finally:
        _lib.set_math('fp32')


def test_pair_loss_and_pair_head_are_bit_identical_to_the_per_half_calls(dev):
    """Round 6 launch diet of the paired discriminator pass: iprgan_loss_pair_* (both hinge terms + their sum, one launch each way)
    and iprgan_gemv_*_pair (the GEMV head of both half-batches, each with its own sigma, one launch per kernel) must reproduce,
    bit for bit, the per-half calls they replace."""
    from iprgan import _lib, ops, tools
    _lib.set_math('fp32x3')
    try:
        B, K = 24, 2048
        x = ops.to_kind(rnd(B, K, seed=1).to(dev), ops.ST_X3)
        w, bias = rnd(K, seed=2, scale=K ** -0.5).to(dev), rnd(1, seed=3).to(dev)
        s0, s1 = torch.tensor([1.7], device=dev), torch.tensor([0.6], device=dev)
        y = ops.gemv_fwd_pair(x, w, bias, s0, s1)
        ya, yb = ops.gemv_fwd(x[:B // 2], w, bias, s0), ops.gemv_fwd(x[B // 2:], w, bias, s1)
        assert torch.equal(y, torch.cat([ya, yb]))
        dy = rnd(B, seed=4).to(dev)
        dx, dw2, db2 = ops.gemv_bwd_pair(x, w, dy, s0, s1, True, True, x, _lib.ACT_LRELU, 0.1)
        ref = ops._empty_like(x)
        halves = []
        for h, (sl, sg) in enumerate(((slice(0, B // 2), s0), (slice(B // 2, None), s1))):
            _, dwp, db = ops.gemv_bwd(x[sl], w, dy[sl].contiguous(), sg, True, True, x[sl], _lib.ACT_LRELU, 0.1, dx_out=ref[sl])
            halves.append((dwp, db))
        assert torch.equal(ops.f32(dx), ops.f32(ref))
        for h in (0, 1):
            assert torch.equal(dw2[h], halves[h][0]) and torch.equal(db2[h:h + 1], halves[h][1])
        # losses: hinge(real) over the first half, hinge(fake) over the second, their sum; gradient of the sum
        logits = (rnd(2 * 100, seed=5) * 2).to(dev).requires_grad_()
        la, lb, ls = tools.loss_pair(_lib.LOSS_HINGE_REAL, _lib.LOSS_HINGE_FAKE, logits, 100)
        ls.backward(torch.tensor(0.7, device=dev))
        l2 = logits.detach().clone().requires_grad_()
        ra, rb = tools.loss_value(_lib.LOSS_HINGE_REAL, l2[:100]), tools.loss_value(_lib.LOSS_HINGE_FAKE, l2[100:])
        (ra + rb).backward(torch.tensor(0.7, device=dev))
        assert torch.equal(la, ra) and torch.equal(lb, rb) and torch.equal(ls, ra + rb)
        assert torch.equal(logits.grad, l2.grad)
        with pytest.raises(RuntimeError, match='per half'):
            ops.loss_pair_fwd(_lib.LOSS_HINGE_REAL, _lib.LOSS_HINGE_FAKE, torch.zeros(1024, device=dev), 512)
    finally:
        _lib.set_math('fp32')
