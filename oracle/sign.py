"""Oracle sign-loss watermark (reference tools/sign_model.py:6-60).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Integer parts (bit stream, signs,
bit-error count) are restated in numpy/pure Python and must match bit-exactly.
"""
import random

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

NORM_TYPES = (nn.BatchNorm2d, nn.InstanceNorm2d)


def string_bits(string):
    """Bits of ``string + '\\t'``, 8 per character, MSB first (sign_model.py:11-12)."""
    raw = (string + '\t').encode('latin-1')      # ord(c) < 256 for the configs' ASCII strings
    return np.unpackbits(np.frombuffer(raw, dtype=np.uint8)).astype(np.int64)


class BitStream:
    """Cycling bit source whose cursor persists across layers (sign_model.py:14-23)."""

    def __init__(self, string=None):
        self.bits = None if string is None else string_bits(string)
        self.cursor = 0

    def take(self, n):
        if self.bits is None:                    # sign_model.py:16-17
            return [random.randint(0, 1) for _ in range(n)]
        idx = (self.cursor + np.arange(n)) % len(self.bits)
        self.cursor += n
        return self.bits[idx].tolist()


def norm_layers(model):
    """(safe_name, module) for every BN / IN layer in named_modules() order
    (sign_model.py:34-36)."""
    for name, m in model.named_modules():
        if isinstance(m, NORM_TYPES):
            yield name.replace('.', '_'), m


class SignLossModel(nn.Module):
    def __init__(self, model, config, **kwargs):
        super().__init__()
        self.gamma_0 = config.gamma_0
        self.stream = BitStream(config.string)
        for safe, m in norm_layers(model):       # sign_model.py:33-40
            sign = torch.tensor(self.stream.take(m.weight.size(0)), dtype=torch.float32) * 2 - 1
            m.weight.data.abs_().mul_(sign.to(m.weight.device))
            self.register_buffer(safe, sign)

    def forward(self, model):                    # sign_model.py:42-49
        total = 0
        for safe, m in norm_layers(model):
            total = total + F.relu(self.gamma_0 - m.weight * getattr(self, safe)).mean()
        return total

    def compute_ber(self, model):                # sign_model.py:51-60
        wrong, count = 0, 0
        for safe, m in norm_layers(model):
            sign = getattr(self, safe)
            wrong = wrong + (m.weight.sign() != sign).float().sum()
            count += sign.size(0)
        return wrong / count
