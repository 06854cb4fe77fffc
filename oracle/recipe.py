"""Deterministic, platform-independent tensors for parity tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Network weights are far too large
to commit as fixtures (ConvGenerator64 alone is 28 MB), so golden vectors store
only inputs/outputs/summaries and both sides rebuild the weights from this recipe
(numpy PCG64 streams are bit-identical across machines).
"""
import numpy as np
import torch


def _rng(seed, i):
    return np.random.default_rng([int(seed), int(i)])


def tensor(seed, i, shape, scale=1.0, shift=0.0, dist='normal'):
    g = _rng(seed, i)
    a = g.standard_normal(shape) if dist == 'normal' else g.random(shape)
    return torch.from_numpy((a * scale + shift).astype(np.float32))


def state_for(module, seed):
    """A full state_dict for ``module`` (keys/shapes taken from the module)."""
    out = {}
    for i, (key, ref) in enumerate(module.state_dict().items()):
        leaf = key.rsplit('.', 1)[-1]
        shape = tuple(ref.shape)
        if leaf == 'num_batches_tracked':
            t = torch.zeros(shape, dtype=ref.dtype)
        elif leaf == 'running_mean':
            t = tensor(seed, i, shape, 0.1)
        elif leaf == 'running_var':
            t = tensor(seed, i, shape, 0.2, 1.0, dist='uniform')
        elif leaf in ('weight_u', 'weight_v'):
            t = tensor(seed, i, shape)
            t = t / t.norm()
        elif leaf == 'bias':
            t = tensor(seed, i, shape, 0.1)
        elif ref.dim() == 1 and ref.numel() == 1:          # PReLU slope
            t = tensor(seed, i, shape, 0.05, 0.25)
        elif ref.dim() == 1:                               # norm gamma: some |g| < gamma_0, some < 0
            t = tensor(seed, i, shape, 0.5, 0.3)
        else:                                              # conv / conv-transpose / linear weight
            fan = max(1, ref.numel() // shape[0])
            t = tensor(seed, i, shape, float(np.sqrt(2.0 / fan)))
        out[key] = t
    return out


def fill(module, seed):
    module.load_state_dict(state_for(module, seed))
    return module


def summary(t, n_head=8, n_stride=64, n_dense=2048):
    """Compact fingerprint of a tensor: sum, abs-sum, L2 norm, head, a coarse strided sample (``samp``: 64 values, float64)
    and a dense one (``dense``: up to 2048 values, float32 - small tensors are stored whole).  sum / asum / samp alone would
    pass a defect confined to unsampled entries that preserves the sums; the L2 norm moves with ANY change of the values
    and the dense sample covers every tensor of up to 2048 elements completely."""
    f = t.detach().double().flatten()
    step = max(1, f.numel() // n_stride)
    dstep = max(1, -(-f.numel() // n_dense))
    return {
        'sum': np.float64(f.sum().item()),
        'asum': np.float64(f.abs().sum().item()),
        'l2': np.float64(f.pow(2).sum().sqrt().item()),
        'head': f[:n_head].numpy().astype(np.float64),
        'samp': f[::step][:n_stride].numpy().astype(np.float64),
        'dense': f[::dstep][:n_dense].numpy().astype(np.float32),
    }


def summary_close(t, ref, rtol, atol, what=''):
    """assert that tensor ``t`` matches a stored summary() within tolerance."""
    got = summary(t)
    scale = max(1.0, float(ref['asum'])) / max(1, t.numel()) ** 0.5
    assert abs(got['sum'] - float(ref['sum'])) <= atol * t.numel() ** 0.5 + rtol * float(ref['asum']) + rtol * scale, \
        f'{what}: sum {got["sum"]} vs {float(ref["sum"])}'
    assert abs(got['asum'] - float(ref['asum'])) <= rtol * float(ref['asum']) + atol * t.numel(), \
        f'{what}: asum {got["asum"]} vs {float(ref["asum"])}'
    np.testing.assert_allclose(got['head'], ref['head'], rtol=rtol, atol=atol, err_msg=f'{what}: head')
    np.testing.assert_allclose(got['samp'], ref['samp'], rtol=rtol, atol=atol, err_msg=f'{what}: samp')
    if 'l2' in ref:
        assert abs(got['l2'] - float(ref['l2'])) <= rtol * float(ref['l2']) + atol * t.numel() ** 0.5, f'{what}: l2'
        np.testing.assert_allclose(got['dense'], ref['dense'], rtol=rtol, atol=atol + 1e-6, err_msg=f'{what}: dense')


def pack_summary(prefix, t, into):
    for k, v in summary(t).items():
        into[f'{prefix}::{k}'] = v


def unpack_summary(prefix, z):
    return {k: z[f'{prefix}::{k}'] for k in ('sum', 'asum', 'l2', 'head', 'samp', 'dense') if f'{prefix}::{k}' in z}
