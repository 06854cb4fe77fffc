"""pytest config: registers the `gpu` marker and puts the repo + package on sys.path.

`-m "not gpu"` runs on the CPU-only build container (oracle vs golden vectors, host logic,
C-ABI symbol table, gloo world_size-2); `-m gpu` runs on a real MI355X through the C-ABI.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'ipr-gan_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
sys.dont_write_bytecode = True
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
    return load
