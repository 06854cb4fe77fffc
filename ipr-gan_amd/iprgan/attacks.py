"""Removal attacks of the reference's evaluation scripts, as library functions.

``sign_flip_`` is the loop body of sign_flip.py:59-75, ``prune_`` the one of prune.py:46-57; both act on the
generator whose BatchNorm / InstanceNorm scales carry the watermark, and are followed in the reference by
``SignLossModel.compute_ber`` (experiments/image_generation.py:161-164) - here the exact int64 count kernel
``iprgan_sign_ber``.  ``matching_p_value`` is tools/phash_pvalue.py:34-37 (the hash itself is the third-party
pdqhash, absent offline: only the binomial tail is restated)."""
import numpy as np
import torch
import torch.nn as nn

__all__ = ['norm_scales', 'sign_flip_', 'prune_', 'matching_p_value']

_NORMS = (nn.BatchNorm2d, nn.InstanceNorm2d)


def norm_scales(net):
    """The watermark carriers in ``modules()`` order (sign_flip.py:61-63 counts exactly these)."""
    return [m.weight for m in net.modules() if isinstance(m, _NORMS) and m.weight is not None]


@torch.no_grad()
def sign_flip_(net, percent, generator=None):
    """Negates ``int(n * percent / 100)`` randomly chosen norm scales in place; returns the +-1 mask used."""
    scales = norm_scales(net)
    n = sum(w.numel() for w in scales)
    nflip = int(n * percent / 100)
    mask = torch.ones(n)
    mask[torch.randperm(n, generator=generator)[:nflip]] *= -1
    rest = mask
    for w in scales:
        k = w.numel()
        w.mul_(rest[:k].to(w))
        rest = rest[k:]
    return mask


@torch.no_grad()
def prune_(state_dict, percent):
    """Zeroes, in place, every entry of a network's state_dict whose magnitude is below the ``percent``-th
    percentile of ALL entries (buffers included, as prune.py:47-53 does); returns the threshold."""
    flat = np.concatenate([v.detach().abs().cpu().double().numpy().ravel() for v in state_dict.values()])
    threshold = np.percentile(flat, percent)
    for v in state_dict.values():
        v[v.abs() < threshold] = 0
    return float(threshold)


def matching_p_value(hash_x, hash_y):
    """P(at least r of n fair bits agree), r = n - hamming distance, per row of two boolean hash matrices."""
    from scipy.stats import binom
    hx, hy = np.asarray(hash_x, dtype=bool), np.asarray(hash_y, dtype=bool)
    n = hx.shape[1]
    r = n - (hx ^ hy).sum(axis=1)
    return torch.tensor([1.0 - binom(n=n, p=0.5).cdf(ri - 1) for ri in r], dtype=torch.float32)
