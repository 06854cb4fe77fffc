"""GPU parity of math mode 'fp32x3' (include/iprgan.h: IPRGAN_MATH_FP32X3): fp32 values stored as three exact bf16 planes
(x = h + m + l, IPRGAN_ST_X3; produced by the conv / norm / activation epilogues and weight prep), a product block accumulated
from six bf16 MFMAs into two fp32 accumulators (conv_x3.hip, wgrad_x3.hip, and the SPLIT forms of conv_igemm.hip).

The claim under test is "fp32 arithmetic", so every check here uses the fp32 tolerances:
  * per layer, the distance to a float64 convolution must not exceed the exact-fp32 MFMA's (forward, backward-data,
    backward-weight; every register-staged, ring, halo and 16x16x32 tile that applies to the layer);
  * split / join of the planes is exact;
  * the network / training-step parity tests of test_gpu_models.py with every 'fp32' request routed to this mode.
    (test_gpu_models.py and the conv tests of test_gpu_ops.py are themselves collected in both modes by default -
    conftest.both_math_modes; the re-runs here predate that and stay as a second entry point.)"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import test_gpu_models as T

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


@pytest.fixture
def via_x3():
    """Every 'fp32' request becomes 'fp32x3' for the duration of one test."""
    from iprgan import _lib
    _lib._FP32_VIA_X3 = True
    _lib.set_math('fp32')
    assert _lib.get_math() == 'fp32x3'
    try:
        yield
    finally:
        _lib._FP32_VIA_X3 = False
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')
        assert _lib.get_math() == 'fp32'


LAYERS = [  # B, cin, cout, k, stride, pad, H, transposed
    (8, 64, 128, 3, 1, 1, 32, False), (8, 128, 128, 4, 2, 1, 32, False), (4, 256, 512, 3, 1, 1, 8, False),
    (5, 64, 64, 3, 1, 1, 8, False), (3, 128, 64, 4, 2, 1, 8, True),       # 8x8 maps, ragged batch: four maps per halo tile
    (8, 64, 64, 4, 2, 1, 64, False), (8, 256, 128, 4, 2, 1, 16, True), (2, 128, 256, 3, 1, 1, 24, False),
    (2, 256, 256, 3, 2, 1, 24, False), (2, 512, 1024, 6, 1, 0, 6, False), (4, 96, 160, 3, 1, 1, 20, False),
]


@pytest.mark.parametrize('layer', LAYERS, ids=lambda l: f'B{l[0]}_{l[1]}to{l[2]}_k{l[3]}s{l[4]}_{l[6]}' + ('T' if l[7] else ''))
@pytest.mark.parametrize('tile', [-1, 0, 1, 2, 3, 4, 5, 6, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37])
def test_split_tiles_are_as_close_to_float64_as_the_fp32_tiles(layer, tile, dev):
    """rms error against the float64 convolution, relative to the result's rms: fp32x3 within 1.25x of the fp32 mode's
    own error + 1e-7, and below 1.5e-6 in absolute terms, for y, dx and dw.  (Measured: 0.6x .. 1.0x - the split products
    are exact where the fp32 MFMA rounds every fused multiply-add.)  Tiles 0-6: the split tiles; 8, 11: LDS-DMA ring
    tiles (fp32 tensors only: not candidates for three-plane operands, the forced tile falls back); 18-25: the three-plane
    ring tiles of conv_x3.hip; 26, 27: their halo form (stride-1 gathers: k3 s1 and the 2x2-tap phases of k4 s2 - other
    geometries fall back); 28-31: the ring tiles on the 16x16x32 MFMA; 32-37: the ring with dedicated loader waves (round 5; 35-37 on the 16x16x32 MFMA).  96 -> 160 channels: a multiple of 32 but not of 64
    or 128, so tiles have ragged columns."""
    from iprgan import ops, _lib
    B, cin, cout, k, s, p, H, tr = layer
    g = torch.Generator().manual_seed(1234 + cin + cout)
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
    OH, OW = spec.out_hw(H, H)
    x = (torch.randn(B, H, H, cin, generator=g) + 0.5).to(dev)
    dy = torch.randn(B, OH, OW, cout, generator=g).to(dev)
    w = (torch.randn(*((cin, cout, k, k) if tr else (cout, cin, k, k)), generator=g) * 0.05).to(dev)
    x64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().cpu().requires_grad_(True)
    y64 = F.conv_transpose2d(x64, w64, None, s, p) if tr else F.conv2d(x64, w64, None, s, p)
    y64.backward(dy.double().cpu().permute(0, 3, 1, 2))
    ref = (y64.detach().permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1), w64.grad)
    err = {}
    try:
        for mode in ('fp32', 'fp32x3'):
            _lib.set_math(mode)
            _lib.call('iprgan_debug_force_tiles', tile, -1)
            d = spec.desc(B, H, H)          # (fp32x3: x, y live as three bf16 planes - the storage kinds are part of the descriptor)
            assert (d.x_bf16, d.y_bf16) == ((2, 2) if mode == 'fp32x3' else (0, 0))
            xk, dyk = ops.to_kind(x, d.x_bf16), ops.to_kind(dy, d.y_bf16)
            wf, wb = ops.conv_prep(spec, d, w, None, True, True)
            y = ops.conv_fwd(spec, d, xk, wf, None)
            dx = ops.conv_bwd_data(spec, d, dyk, wb)
            dw = ops.conv_bwd_weight(spec, d, xk, dyk, tuple(w.shape), False)
            dw = dw[0] if isinstance(dw, tuple) else dw
            err[mode] = [float((ops.f32(got).double().cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
                         for got, want in zip((y, dx, dw), ref)]
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')
    for name, e32, ex3 in zip(('y', 'dx', 'dw'), err['fp32'], err['fp32x3']):
        assert ex3 <= e32 + 2e-8 and ex3 < 1.5e-6, f'{name}: fp32x3 {ex3:.3e} vs fp32 {e32:.3e} (rms, against float64)'


@pytest.mark.parametrize('tile', [0, 2, 4, 6, 18, 19, 21, 23, 26, 27, 28, 30, 32, 33, 34, 35, 36, 37])
def test_split_tiles_epilogue_statistics(tile, dev):
    """Column statistics from the epilogue of the split tiles (the BatchNorm that follows takes them instead of a pass
    over y): mean and 1/std against the float64 statistics of the stored y, ragged last tile included (M = 1152)."""
    from iprgan import ops, _lib
    g = torch.Generator().manual_seed(5)
    try:
        _lib.set_math('fp32x3')
        for (B, cin, cout, k, s, p, H) in [(2, 128, 256, 3, 1, 1, 24), (2, 128, 128, 3, 2, 1, 48), (3, 64, 192, 3, 1, 1, 20)]:
            spec = ops.ConvSpec(cin, cout, k, s, p, 0, False)
            d = spec.desc(B, H, H)
            x = ops.to_kind(torch.randn(B, H, H, cin, generator=g).abs().to(dev), d.x_bf16)
            w = (torch.randn(cout, cin, k, k, generator=g) * 0.05).to(dev)
            bias = torch.randn(cout, generator=g).to(dev)
            _lib.call('iprgan_debug_force_tiles', tile, -1)
            wf, _ = ops.conv_prep(spec, d, w, None, True, False)
            y, stats = ops.conv_fwd(spec, d, x, wf, bias, stats=True)
            _, mean, invstd = ops.bn_fwd(y, None, None, None, None, 1e-5, 0.0, True, 0, conv_stats=stats, conv_bias=bias)
            y64 = ops.f32(y).double().reshape(-1, y.shape[-1])
            m64, v64 = y64.mean(0), y64.var(0, unbiased=False)
            assert float(((mean.double() - m64).abs() / v64.sqrt()).max()) < 2e-6
            assert float(((invstd.double() - 1 / (v64 + 1e-5).sqrt()).abs() * (v64 + 1e-5).sqrt()).max()) < 5e-6
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


def test_split_wgrad_candidates_vs_float64(dev):
    """The backward-weight tiles that exist in split form (128x128, 4 and 8 waves; zero and reflect padding) forced one by
    one against float64."""
    from iprgan import ops, _lib
    g = torch.Generator().manual_seed(77)
    try:
        for pad_mode in (0, 1):
            spec = ops.ConvSpec(128, 128, 3, 1, 1, 0, False, pad_mode=pad_mode)
            d = spec.desc(4, 16, 16)
            x = torch.randn(4, 16, 16, 128, generator=g).to(dev)
            dy = torch.randn(4, 16, 16, 128, generator=g).to(dev)
            d0 = d
            x64 = x.double().cpu().permute(0, 3, 1, 2)
            if pad_mode:
                x64 = F.pad(x64, (1, 1, 1, 1), mode='reflect')
            w64 = torch.zeros(128, 128, 3, 3, dtype=torch.float64, requires_grad=True)
            F.conv2d(x64, w64, None, 1, 0 if pad_mode else 1).backward(dy.double().cpu().permute(0, 3, 1, 2))
            want = w64.grad
            seen = 0
            _lib.set_math('fp32x3')
            for cand in range(0, 60):
                _lib.call('iprgan_debug_force_tiles', -1, cand)
                # (cand even: three-plane tensors - candidates other than the 128x128 split tiles fall back to those; odd:
                # the fp32 tensors of a layer the planes rule does not cover)
                planes = cand % 2 == 0
                d = spec.desc(4, 16, 16) if planes else d0
                dw = ops.conv_bwd_weight(spec, d, ops.to_kind(x, 2) if planes else x, ops.to_kind(dy, 2) if planes else dy,
                                         (128, 128, 3, 3), False)
                dw = dw[0] if isinstance(dw, tuple) else dw
                e = float((dw.double().cpu() - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt())
                assert e < 1e-6, f'pad_mode {pad_mode} wgrad candidate {cand}: {e:.3e}'
                seen += 1
            assert seen == 60
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


X3H_SHAPES = [  # B, cin, cout, k, stride, pad, H, W, transposed, reflect
    (4, 64, 128, 3, 1, 1, 16, 16, False, False), (3, 128, 64, 3, 1, 1, 20, 12, False, False),
    (2, 128, 128, 3, 1, 1, 16, 24, False, True), (4, 64, 64, 4, 2, 1, 32, 32, False, False),
    (3, 128, 192, 4, 2, 1, 20, 28, False, False), (4, 128, 64, 4, 2, 1, 8, 8, True, False),
    (2, 256, 128, 4, 2, 1, 6, 10, True, False),
    # k3 s2 (round 6): Discriminator96 / ResnetGenerator downsampling, ragged odd maps, ConvTranspose with and without output padding
    (4, 64, 128, 3, 2, 1, 32, 32, False, False), (3, 128, 64, 3, 2, 1, 21, 13, False, False),
    (2, 256, 128, 3, 2, 1, 8, 12, True, False), (3, 128, 64, 3, 2, 1, 9, 7, True, False, 1),
]


@pytest.mark.parametrize('cand', [-1, 0, 74, 75, 76, 77, 78])
@pytest.mark.parametrize('shape', X3H_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_wgrad_halo_for_three_plane_tensors(shape, cand, dev):
    """Backward-weight of three-plane tensors: the halo form (csrc/wgrad_x3.hip, candidates 74-76; k3 s1 and k4 s2, Conv2d
    and ConvTranspose2d, zero and reflection padding, ragged patches) and the split-M tile that reads the planes directly
    (candidate 0), against float64 next to the fp32 mode's own distance; and on a batch slice of a larger tensor (the
    paired discriminator pass hands such slices: plane stride of the whole tensor)."""
    from iprgan import ops, _lib
    B, cin, cout, k, s, p, H, W, tr, refl = shape[:10]
    op = shape[10] if len(shape) > 10 else 0                 # output_padding of a ConvTranspose2d (resnet_generator.py:27-30)
    g = torch.Generator().manual_seed(99 + cin + cout + H)
    spec = ops.ConvSpec(cin, cout, k, s, p, op, tr, pad_mode=1 if refl else 0)
    OH, OW = spec.out_hw(H, W)
    x = torch.randn(B, H, W, cin, generator=g).to(dev)
    dy = torch.randn(B, OH, OW, cout, generator=g).to(dev)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    x64 = x.double().cpu().permute(0, 3, 1, 2)
    if refl:
        x64 = F.pad(x64, (p, p, p, p), mode='reflect')
    w64 = torch.zeros(*wshape, dtype=torch.float64, requires_grad=True)
    y64 = F.conv_transpose2d(x64, w64, None, s, p, op) if tr else F.conv2d(x64, w64, None, s, 0 if refl else p)
    y64.backward(dy.double().cpu().permute(0, 3, 1, 2))
    want = w64.grad
    rel = lambda got, ref: float((got.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())      # noqa: E731
    try:
        _lib.set_math('fp32')
        e32 = rel(ops.conv_bwd_weight(spec, spec.desc(B, H, W), x, dy, wshape, False)[0], want)
        _lib.set_math('fp32x3')
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        d = spec.desc(B, H, W)
        xp, dyp = ops.to_kind(x, 2), ops.to_kind(dy, 2)
        ex3 = rel(ops.conv_bwd_weight(spec, d, xp, dyp, wshape, False)[0], want)
        assert ex3 <= e32 + 2e-8, f'cand {cand}: fp32x3 {ex3:.3e} vs fp32 {e32:.3e}'
        # the first B - 1 samples as a slice of the B-sample tensors (plane stride of the whole tensor)
        if B > 2:
            w64.grad = None
            y2 = F.conv_transpose2d(x64[:B - 1], w64, None, s, p, op) if tr else F.conv2d(x64[:B - 1], w64, None, s, 0 if refl else p)
            y2.backward(dy[:B - 1].double().cpu().permute(0, 3, 1, 2))
            es = rel(ops.conv_bwd_weight(spec, spec.desc(B - 1, H, W), xp[:B - 1], dyp[:B - 1], wshape, False)[0], w64.grad)
            assert es < 4e-7, f'cand {cand}, batch slice: {es:.3e}'
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


@pytest.mark.parametrize('shape', [(6, 12, 10, 64), (3, 16, 16, 96), (2, 9, 7, 256)], ids=lambda c: 'x'.join(map(str, c)))
@pytest.mark.parametrize('layer', ['bn', 'bn_prelu', 'in'])
def test_norm_layers_take_an_fp32_input_next_to_three_plane_tensors(layer, shape, dev):
    """Storage kind IPRGAN_ST_X3_XF32 of the norm entry points: the layer's input x stays fp32 (the convolution in front of it
    writes 4 instead of 6 bytes per element) while y, the residual, dy and dx are three-plane tensors.  Same arithmetic,
    same summation order as the all-fp32 kernels: forward output, statistics, dx, dgamma, dbeta (and the PReLU slope
    gradient) must be BIT-identical to the fp32 mode's."""
    from iprgan import ops, _lib
    B, H, W, C_ = shape
    g = torch.Generator().manual_seed(5 + C_ + H)
    x = (torch.randn(B, H, W, C_, generator=g) * 1.5 + 0.3).to(dev)
    res = torch.randn(B, H, W, C_, generator=g).to(dev)
    dy = torch.randn(B, H, W, C_, generator=g).to(dev)
    gamma, beta = (torch.rand(C_, generator=g) + 0.5).to(dev), torch.randn(C_, generator=g).to(dev)
    slope = torch.tensor([0.25], device=dev)

    def run(planes):
        rm, rv = torch.zeros(C_, device=dev), torch.ones(C_, device=dev)
        r, d_ = (ops.to_kind(res, 2), ops.to_kind(dy, 2)) if planes else (res, dy)
        if layer == 'bn':
            y, mean, invstd = ops.bn_fwd(x, gamma, beta, rm, rv, 1e-5, 0.1, True, _lib.ACT_RELU, residual=r)
            out = ops.bn_bwd(x, y, d_, gamma, mean, invstd, _lib.ACT_RELU, beta=beta)
        elif layer == 'bn_prelu':
            y, mean, invstd = ops.bn_prelu_fwd(x, gamma, beta, rm, rv, 1e-5, 0.1, True, slope, residual=r)
            out = ops.bn_prelu_bwd(x, d_, gamma, beta, mean, invstd, slope)
        else:
            y, mean, invstd = ops.instnorm_fwd(x, gamma, beta, 1e-5, _lib.ACT_RELU, residual=r)
            out = ops.instnorm_bwd(x, y, d_, gamma, mean, invstd, _lib.ACT_RELU, beta=beta)
        if planes:
            assert ops.is16(y) == 2 and ops.is16(out[0]) == 2 and ops.is16(x) == 0
        return [ops.f32(y), mean, invstd, rm, rv, ops.f32(out[0])] + list(out[1:])
    try:
        _lib.set_math('fp32')
        want = run(False)
        _lib.set_math('fp32x3')
        got = run(True)
    finally:
        _lib.set_math('fp32')
    for i, (a_, b_) in enumerate(zip(got, want)):
        assert torch.equal(a_, b_), f'output {i}: max diff {float((a_ - b_).abs().max()):.3e}'


RGB_SHAPES = [  # B, cin, cout, k, stride, pad, H, W, transposed, reflect: one side RGB (fp32), the other 64 channels (planes)
    (6, 3, 64, 3, 1, 1, 24, 20, False, False),      # SNDiscriminator stem (sn_discriminator.py:9): dy is the three-plane side
    (4, 64, 3, 3, 1, 1, 16, 24, True, False),       # ConvGenerator head (conv_generator.py:21): x is
    (2, 64, 3, 7, 1, 3, 20, 18, False, True),       # ResnetGenerator head (resnet_generator.py:43-44): role-swapped, padded fp32 copy of x
    (2, 3, 64, 7, 1, 3, 18, 20, False, True),       # ResnetGenerator stem (resnet_generator.py:6-7): reflection in the gather
    (3, 64, 3, 9, 1, 4, 16, 16, False, False),      # SRResNet head (sr_resnet.py:17)
]


@pytest.mark.parametrize('cand', [-1, 0, 1])
@pytest.mark.parametrize('shape', RGB_SHAPES, ids=lambda c: '-'.join(map(str, c)))
def test_wgrad_of_rgb_layers_reads_the_three_plane_side_directly(shape, cand, dev):
    """Backward-weight of RGB stems / heads in 'fp32x3' mode: the 64-channel operand is a three-plane tensor, the image side
    fp32; the fp32 tiles sum the planes h + (m + l) - exactly - as they load them (WGradArgs p16 / q16 = 2; the reflect-padded
    role-swapped form joins while it pads), so no fp32 copy of the large tensor is made.  The result must equal, bit for
    bit, what the same candidate computes from the joined fp32 tensor, and sit at the fp32 distance from float64."""
    import ctypes as C
    from iprgan import ops, _lib
    B, cin, cout, k, s, p, H, W, tr, refl = shape
    g = torch.Generator().manual_seed(7 + cin + cout + H + k)
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=1 if refl else 0)
    OH, OW = spec.out_hw(H, W)
    x = torch.randn(B, H, W, ops.c4(cin), generator=g).to(dev)
    dy = torch.randn(B, OH, OW, ops.c4(cout), generator=g).to(dev)
    x[..., cin:] = 0
    dy[..., cout:] = 0
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    x64 = x[..., :cin].double().cpu().permute(0, 3, 1, 2)
    if refl:
        x64 = F.pad(x64, (p, p, p, p), mode='reflect')
    w64 = torch.zeros(*wshape, dtype=torch.float64, requires_grad=True)
    y64 = F.conv_transpose2d(x64, w64, None, s, p) if tr else F.conv2d(x64, w64, None, s, 0 if refl else p)
    y64.backward(dy[..., :cout].double().cpu().permute(0, 3, 1, 2))
    want = w64.grad
    rel = lambda got, ref: float((got.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())      # noqa: E731
    try:
        # the same values as fp32 tensors on both sides through the same candidate (the exact fp32 tiles of mode 'fp32')
        _lib.set_math('fp32')
        _lib.call('iprgan_debug_force_tiles', -1, cand)
        ref32, db32 = ops.conv_bwd_weight(spec, spec.desc(B, H, W), x, dy, wshape, True)
        _lib.set_math('fp32x3')
        d = spec.desc(B, H, W)
        assert sorted((d.x_bf16, d.y_bf16)) == [0, 2]
        assert _lib.query('iprgan_conv_wgrad_takes_bf16', C.byref(d)) == 1
        xk, dyk = ops.to_kind(x, d.x_bf16), ops.to_kind(dy, d.y_bf16)
        got, db = ops.conv_bwd_weight(spec, d, xk, dyk, wshape, True)
        assert cand < 0 or torch.equal(got, ref32), f'cand {cand}: differs from the joined operands by {float((got - ref32).abs().max()):.3e}'
        assert rel(got, want) < 4e-7
        dy64 = dy[..., :cout].double().cpu()
        tol = 1e-6 * float(dy64.abs().sum((0, 1, 2)).max())          # (a sum of B * OH * OW signed terms: cancellation)
        assert float((db.double().cpu() - dy64.sum((0, 1, 2))).abs().max()) <= tol
        assert float((db.double().cpu() - db32.double().cpu()).abs().max()) <= tol
    finally:
        _lib.call('iprgan_debug_force_tiles', -1, -1)
        _lib.set_math('fp32')


@pytest.mark.parametrize('name', [n for n in T.HIP_NETS if n != 'Discriminator96'])
def test_networks_vs_reference_golden_through_split_tiles(name, golden, dev, via_x3):
    T.test_net_vs_reference_golden(name, golden, dev)


@pytest.mark.parametrize('name', ['ConvDiscriminator', 'Resnet6Blocks', 'SNDiscriminator64', 'ConvGenerator64', 'SRResNet'])
def test_network_accuracy_against_float64_through_split_tiles(name, dev, via_x3):
    T.test_net_accuracy_against_float64(name, dev)


@pytest.mark.parametrize('wbox', [True, False])
def test_dcgan_steps_vs_reference_golden_through_split_tiles(wbox, golden, dev, via_x3):
    T.test_dcgan_steps_vs_reference_golden(wbox, golden, dev)


def test_training_steps_vs_reference_golden_through_split_tiles(golden, dev, via_x3):
    """DCGAN at batch 128 against the live oracle, DCGAN-128 (two steps), SRGAN (pretrain + GAN step) and CycleGAN (two
    steps) against the fixtures from the real reference, with the fp32 tolerances of those tests."""
    T.test_dcgan_bs128_step_vs_oracle(dev)
    T.test_dcgan128_steps_vs_reference_golden(golden, dev)
    T.test_srgan_steps_vs_reference_golden(golden, dev)
    T.test_cyclegan_steps_vs_reference_golden(golden, dev)


def test_split_step_is_deterministic_captures_and_keeps_the_watermark(dev, via_x3):
    """Full-size DCGAN-64 step (batch 128): two runs from the same state are bit-identical, the sign-loss watermark reads
    back with BER 0; and the step captured in a HIP graph is bit-identical to the eager one in this mode too."""
    T.test_full_size_step_is_deterministic_and_keeps_watermark(dev)
    T._graphed_vs_eager(dev, 'fp32', 5, 3, 2)


def test_bench_roofline_of_split_mode_is_priced_against_a_sixth_of_the_bf16_peak():
    import bench
    assert bench.PEAK_X3_MFMA == pytest.approx(bench.PEAK_BF16_MFMA / 6)
    assert np.isclose(bench.PEAK_X3_MFMA / 1e12, 416.7, atol=0.1)


# ---- the ends of fp32's exponent range (VERDICT r04 next #8a; the contract is written down in include/iprgan.h next to
# IPRGAN_ST_X3): where "x = h + (m + l) exactly" holds, what happens below and above it, and that a convolution on
# operands out there is still as close to float64 as the fp32 MFMA's.
def _round_trip(x):
    """fp32 -> three planes -> fp32 through iprgan_cast_planes (a bf16 tensor is a three-plane tensor only in 'fp32x3' mode)."""
    from iprgan import ops, _lib
    try:
        _lib.set_math('fp32x3')
        return ops.f32(ops.to_kind(x, ops.ST_X3))
    finally:
        _lib.set_math('fp32')


def _mantissas(n, seed):
    g = torch.Generator().manual_seed(seed)
    m = 1.0 + torch.randint(0, 1 << 23, (n,), generator=g).double() / float(1 << 23)      # 24 significant bits, all used
    return m * torch.where(torch.rand(n, generator=g) < 0.5, -1.0, 1.0).double()


@pytest.mark.parametrize('exp', [-110, -105, -100, -90, -30, 0, 60, 100, 120, 126])
def test_planes_are_exact_down_to_2_to_the_minus_110_and_up_to_the_bf16_maximum(exp, dev):
    """Every fp32 value with 2^-110 <= |x| <= 0x7F7F0000 (the largest finite bf16, 3.3895e38) round-trips bit for bit:
    the low plane's least significant bit is x's, and bf16 reaches down to 2^-133 (subnormal)."""
    from iprgan import ops
    x = (_mantissas(4096, exp) * 2.0 ** exp).float()
    x = x[x.abs() <= 3.3895313892515355e38]
    assert x.numel() > 2000 and bool((x.double().abs() >= 2.0 ** -110).all())
    xd = x.to(dev)
    back = _round_trip(xd)
    assert torch.equal(back, xd), f'2^{exp}: {int((back != xd).sum())} of {x.numel()} values changed'


@pytest.mark.parametrize('exp', [-111, -118, -126, -130, -140, -149])
def test_planes_below_2_to_the_minus_110_keep_the_bf16_subnormal_grid(exp, dev):
    """Below 2^-110 the planes run out of exponent range before fp32 does: the stored value is x rounded to a multiple of
    2^-133 (bf16's subnormal spacing) - absolute error at most 2^-134 per plane rounding, i.e. 24 significant bits at 2^-110,
    8 at 2^-126.  Nothing becomes non-finite and signs are kept."""
    from iprgan import ops
    x = (_mantissas(2048, 7 - exp) * 2.0 ** exp).float()               # (fp32 subnormals below 2^-126 included)
    xd = x.to(dev)
    back = _round_trip(xd)
    assert bool(torch.isfinite(back).all())
    err = (back.double() - xd.double()).abs().max().item()
    assert err <= 2.0 ** -133, f'2^{exp}: worst absolute error 2^{np.log2(max(err, 1e-300)):.1f}'
    assert bool(((back == 0) | (torch.sign(back) == torch.sign(xd))).all())


def test_planes_never_turn_a_non_finite_or_over_range_value_into_a_finite_one(dev):
    """NaN stays NaN.  +-inf and finite values beyond the largest bf16 (|x| > 0x7F7F7FFF: h rounds to inf, the residual
    planes are inf - inf) come back NON-FINITE (NaN) - never as a finite number, so a diverged run is still detected by
    the finiteness checks that the fp32 mode would trip (bench.py asserts on the metrics; models/base.py has none)."""
    from iprgan import ops
    big = torch.tensor([0x7F7F8000, 0x7F7FFFFF, 0xFF7F8000, 0xFF7FFFFF], dtype=torch.int64).to(torch.int32).view(torch.float32)
    x = torch.cat([torch.tensor([float('inf'), float('-inf'), float('nan')]), big, torch.tensor([1.0, -2.5, 3.3895313892515355e38])]).to(dev)
    x = x.repeat(8)[:64].contiguous()
    back = _round_trip(x)
    fin = torch.isfinite(x) & (x.abs() <= 3.3895313892515355e38)
    assert torch.equal(back[fin], x[fin])
    assert not bool(torch.isfinite(back[~fin]).any()), back[~fin]


@pytest.mark.parametrize('xs,ws', [(-100, 60), (-60, -40), (100, -80), (60, 40), (-110, 100)])
def test_conv_at_the_ends_of_the_exponent_range_is_as_close_to_float64_as_the_fp32_mfma(xs, ws, dev):
    """64 -> 128 k3 with activations at 2^xs and weights at 2^ws (products between 2^-100 and 2^100): relative rms distance to
    the float64 convolution at or below the exact fp32 MFMA's, for y, dx and dw - the planes carry the full mantissa over
    the whole range the product itself survives in."""
    from iprgan import ops, _lib
    g = torch.Generator().manual_seed(99 + xs)
    B, cin, cout, H = 4, 64, 128, 16
    spec = ops.ConvSpec(cin, cout, 3, 1, 1, 0, False)
    x = (torch.randn(B, H, H, cin, generator=g).double() * 2.0 ** xs).float().to(dev)
    dys = -(xs + ws) // 2              # gradients scaled so that dx = dy * w and dw = x * dy stay inside fp32's range too
    dy = (torch.randn(B, H, H, cout, generator=g).double() * 2.0 ** dys).float().to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g).double() * 0.05 * 2.0 ** ws).float().to(dev)
    x64, w64 = x.double().cpu().permute(0, 3, 1, 2).requires_grad_(True), w.double().cpu().requires_grad_(True)
    F.conv2d(x64, w64, None, 1, 1).backward(dy.double().cpu().permute(0, 3, 1, 2))
    with torch.no_grad():
        ref = (F.conv2d(x64, w64, None, 1, 1).permute(0, 2, 3, 1), x64.grad.permute(0, 2, 3, 1), w64.grad)
    err = {}
    try:
        for mode in ('fp32', 'fp32x3'):
            _lib.set_math(mode)
            d = spec.desc(B, H, H)
            xk, dyk = ops.to_kind(x, d.x_bf16), ops.to_kind(dy, d.y_bf16)
            wf, wb = ops.conv_prep(spec, d, w, None, True, True)
            y = ops.conv_fwd(spec, d, xk, wf, None)
            dx = ops.conv_bwd_data(spec, d, dyk, wb)
            dw = ops.conv_bwd_weight(spec, d, xk, dyk, tuple(w.shape), False)
            dw = dw[0] if isinstance(dw, tuple) else dw
            got = [ops.f32(t).double().cpu() for t in (y, dx, dw)]
            assert all(bool(torch.isfinite(t).all()) for t in got)
            err[mode] = [float((a - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()) for a, b in zip(got, ref)]
    finally:
        _lib.set_math('fp32')
    for name, e32, ex3 in zip(('y', 'dx', 'dw'), err['fp32'], err['fp32x3']):
        assert ex3 <= e32 + 2e-8 and ex3 < 1.5e-6, f'{name} at 2^{xs} x 2^{ws}: fp32x3 {ex3:.3e} vs fp32 {e32:.3e}'


def test_conv_under_heavy_cancellation_of_the_leading_terms(dev):
    """The h x h' products - everything the first accumulator of the six-MFMA form holds - cancel EXACTLY (activations +-A in
    bf16-representable pairs against one bf16-representable weight), so the result is made of the five small terms alone:
    x = +-A (1 + d), d ~ 2^-10, lands in the m / l planes.  The distance to float64 relative to the (tiny) result must not
    exceed the exact fp32 MFMA's, which rounds every partial sum at the magnitude of A."""
    from iprgan import ops, _lib
    g = torch.Generator().manual_seed(4242)
    B, cin, cout, H = 2, 64, 64, 12
    spec = ops.ConvSpec(cin, cout, 3, 1, 0, 0, False)             # no padding: every output sums all 576 taps
    sign = torch.tensor([1.0, -1.0]).repeat(cin // 2)               # channel pairs (+A, -A)
    A = 3.0
    delta = (torch.randn(B, H, H, cin, generator=g) * 2.0 ** -10).clamp(-2.0 ** -9, 2.0 ** -9)      # |A d| < half a bf16 ulp of A: h = +-A
    x = (A * sign * (1.0 + delta)).float().to(dev)
    wcol = (torch.randint(1, 8, (cout, 1, 1, 1), generator=g).float() * 0.25)           # bf16-representable, one value per output channel
    w = wcol.expand(cout, cin, 3, 3).contiguous().to(dev)
    x64 = x.double().cpu().permute(0, 3, 1, 2)
    ref = F.conv2d(x64, w.double().cpu(), None, 1, 0).permute(0, 2, 3, 1)
    lead = F.conv2d((A * sign).double().view(1, cin, 1, 1).expand(1, cin, H, H), w.double().cpu(), None, 1, 0)
    assert float(lead.abs().max()) == 0.0                            # the leading terms really cancel
    assert float(ref.abs().mean()) < 0.05 * A                        # ... and what is left is small against A
    err = {}
    try:
        for mode in ('fp32', 'fp32x3'):
            _lib.set_math(mode)
            d = spec.desc(B, H, H)
            wf, _ = ops.conv_prep(spec, d, w, None, True, False)
            y = ops.f32(ops.conv_fwd(spec, d, ops.to_kind(x, d.x_bf16), wf, None)).double().cpu()
            err[mode] = float((y - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    finally:
        _lib.set_math('fp32')
    assert err['fp32x3'] <= err['fp32'] + 2e-8, err


def test_conv_with_a_non_finite_input_gives_non_finite_outputs_exactly_where_fp32_does(dev):
    """One +inf and one NaN element in the activations: every output whose receptive field contains one of them is
    non-finite in BOTH modes (the three-plane mode yields NaN where the fp32 mode yields inf: inf x 0 in the residual
    planes), every other output is finite and equal within the fp32 tolerance."""
    from iprgan import ops, _lib
    g = torch.Generator().manual_seed(11)
    B, cin, cout, H = 1, 64, 64, 16
    spec = ops.ConvSpec(cin, cout, 3, 1, 1, 0, False)
    x = torch.randn(B, H, H, cin, generator=g)
    x[0, 4, 5, 7] = float('inf')
    x[0, 11, 9, 40] = float('nan')
    x = x.to(dev)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(dev)
    out = {}
    try:
        for mode in ('fp32', 'fp32x3'):
            _lib.set_math(mode)
            d = spec.desc(B, H, H)
            wf, _ = ops.conv_prep(spec, d, w, None, True, False)
            out[mode] = ops.f32(ops.conv_fwd(spec, d, ops.to_kind(x, d.x_bf16), wf, None)).cpu()
    finally:
        _lib.set_math('fp32')
    bad = torch.zeros(H, H, dtype=torch.bool)
    bad[3:6, 4:7] = True
    bad[10:13, 8:11] = True
    for mode in out:
        fin = torch.isfinite(out[mode][0]).all(dim=-1)
        assert torch.equal(fin, ~bad), mode
    np.testing.assert_allclose(out['fp32x3'][0][~bad].numpy(), out['fp32'][0][~bad].numpy(), rtol=2e-4, atol=2e-5)

