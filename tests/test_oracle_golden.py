"""Pins the CPU oracle to the golden vectors captured from the real reference
(oracle/gen_golden.py).  CPU only; this is what makes the oracle trustworthy as the
checker for the HIP path."""
import numpy as np
import pytest
import torch

from oracle import cases, gan, nets, recipe, sign

RTOL, ATOL = 2e-4, 2e-5      # same torch CPU kernels; slack only for thread-count dependent summation order


def compare(res, ref, rtol=RTOL, atol=ATOL, policy=None):
    """Every array of the golden file must be matched.  ``policy(key) -> (rtol, atol)`` overrides the
    tolerance per entry ('a/b::sum' style keys are tensor summaries, see oracle/recipe.py)."""
    keys = set(ref.files)
    summ = {k.rsplit('::', 1)[0] for k in keys if '::' in k}

    def tol(k):
        return policy(k) if policy else (rtol, atol)

    for k in sorted(keys):
        if '::' in k:
            continue
        assert k in res, f'missing {k}'
        a, b = np.asarray(res[k]), ref[k]
        if b.dtype.kind in 'iuU':
            assert np.array_equal(a, b), k
        else:
            r, t = tol(k)[:2]
            np.testing.assert_allclose(a, b, rtol=r, atol=t + 2e-5 * float(np.abs(b).max() if b.size else 0.0),
                                       err_msg=k)
    for p in sorted(summ):
        pol = tol(p)
        if len(pol) == 3:          # (rtol, atol, 'scale'): only the overall magnitude is comparable
            a, b = float(res[f'{p}::asum']), float(ref[f'{p}::asum'])
            assert abs(a - b) <= pol[0] * abs(b) + pol[1], f'{p}::asum {a} vs {b}'
            continue
        r, t = pol
        n_est = max(1.0, float(ref[f'{p}::asum']) / max(1e-12, float(np.abs(ref[f'{p}::samp']).mean())))
        for f in ('sum', 'asum', 'head', 'samp'):
            a, b = np.asarray(res[f'{p}::{f}']), ref[f'{p}::{f}']
            if f in ('sum', 'asum'):      # element-wise slack accumulates like a random walk in the sums
                at = t * np.sqrt(n_est) + r * float(ref[f'{p}::asum']) * (1.0 if f == 'sum' else 0.0)
            else:
                # entries far below the tensor's own scale carry that scale's rounding noise
                at = t + 2e-5 * float(np.abs(b).max())
            np.testing.assert_allclose(a, b, rtol=r, atol=at, err_msg=f'{p}::{f}')


@pytest.mark.parametrize('name', list(cases.NET_CASES))
def test_net_matches_reference(name, golden):
    torch.manual_seed(0)
    compare(cases.run_net_case(nets, name), golden('net_' + name))


@pytest.mark.parametrize('name', list(cases.SIGN_CASES))
def test_sign_bits_ber_bit_exact(name, golden):
    res = cases.run_sign_case(nets, sign.SignLossModel, gan.Cfg, name)
    ref = golden('sign_' + name)
    assert np.array_equal(res['signs'], ref['signs'])
    assert res['signs'].size == cases.SIGN_CASES[name]
    assert list(res['names']) == list(ref['names'])
    assert res['ber_clean'] == ref['ber_clean'] == 0.0
    assert res['ber_corrupt'] == ref['ber_corrupt']          # integer count / n: exact
    np.testing.assert_allclose(res['loss_corrupt'], ref['loss_corrupt'], rtol=1e-6)
    np.testing.assert_allclose(res['loss_clean'], ref['loss_clean'], rtol=1e-6)


def test_bitstream_known_answer(golden):
    # 'E'=0x45,'X'=0x58 -> first 16 signs
    bits = sign.BitStream('EXAMPLE A').take(200)
    assert np.array_equal(np.array(bits, dtype=np.int8), golden('bits_EXAMPLE_A')['bits'])
    first = [2 * b - 1 for b in bits[:16]]
    assert first == [-1, 1, -1, -1, -1, 1, -1, 1, -1, 1, -1, 1, 1, -1, -1, -1]
    assert len(sign.string_bits('EXAMPLE A')) == 80     # wraps at 80 bits


@pytest.mark.parametrize('wbox', [True, False])
def test_dcgan_steps_match_reference(wbox, golden):
    torch.manual_seed(0)
    res = cases.run_dcgan_steps(gan.Cfg, gan, gan.CPU, n_steps=3 if wbox else 2, wbox=wbox)
    compare(res, golden('dcgan_steps_wbox' if wbox else 'dcgan_steps_plain'), rtol=5e-4, atol=5e-5)


def test_srgan_steps_match_reference(golden):
    torch.manual_seed(0)
    compare(cases.run_srgan_steps(gan.Cfg, gan, gan.CPU), golden('srgan_steps_wbox'), rtol=5e-4, atol=5e-5)


def test_cyclegan_steps_match_reference(golden):
    torch.manual_seed(0)
    compare(cases.run_cyclegan_steps(gan.Cfg, gan, gan.CPU), golden('cyclegan_steps_wbox'), rtol=5e-4, atol=5e-5)


def test_vae_steps_match_reference(golden):
    """Encoder32 / Decoder32 / models.VAE: KL + BCE, one Adam with weight decay over both
    nets, sign loss on the decoder's BatchNorm; eps of the reparameterisation replayed from the CPU generator."""
    compare(cases.run_vae_steps(gan.Cfg, gan, gan.CPU), golden('vae_steps_wbox'), rtol=5e-4, atol=5e-5)
