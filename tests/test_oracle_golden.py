"""Pins the CPU oracle to the golden vectors captured from the real reference
(oracle/gen_golden.py).  CPU only; this is what makes the oracle trustworthy as the
checker for the HIP path."""
import numpy as np
import pytest
import torch

from oracle import cases, gan, nets, recipe, sign

RTOL, ATOL = 2e-4, 2e-5      # same torch CPU kernels; slack only for thread-count dependent summation order


def compare(res, ref, rtol=RTOL, atol=ATOL, weight_atol=None):
    """weight_atol: absolute slack for post-Adam weights ('final/<net>/...' non-buffer entries).  Adam
    moves every weight by ~lr per step in the direction sign(m)/sqrt(v): where a gradient is at
    rounding-noise level the direction itself is noise, so implementations that agree to 1e-6 in the
    gradients can still differ by up to 2*lr*steps in such a weight."""
    keys = set(ref.files)
    from oracle.cases import BUFFER_LEAVES

    def slack(k):
        base = k.split('::')[0]
        is_w = base.startswith(('final/G/', 'final/D/')) and base.rsplit('.', 1)[-1] not in BUFFER_LEAVES
        return weight_atol if (weight_atol and is_w) else atol
    summ = {k.rsplit('::', 1)[0] for k in keys if '::' in k}
    for k in sorted(keys):
        if '::' in k:
            continue
        assert k in res, f'missing {k}'
        a, b = np.asarray(res[k]), ref[k]
        if b.dtype.kind in 'iuU':
            assert np.array_equal(a, b), k
        else:
            np.testing.assert_allclose(a, b, rtol=rtol, atol=slack(k), err_msg=k)
    for p in sorted(summ):
        for f in ('sum', 'asum', 'head', 'samp'):
            a, b = np.asarray(res[f'{p}::{f}']), ref[f'{p}::{f}']
            scale = float(ref[f'{p}::asum']) if f == 'sum' else 0.0
            at = slack(p)
            if at != atol and f in ('sum', 'asum'):        # per-element slack accumulates in the sums
                at = at * np.sqrt(max(1.0, float(ref[f'{p}::asum']) / max(1e-12, np.abs(ref[f'{p}::samp']).mean())))
            np.testing.assert_allclose(a, b, rtol=rtol, atol=at + rtol * scale, err_msg=f'{p}::{f}')


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
