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


# ---- both math modes: the GPU parity tests of test_gpu_models.py and the convolution tests of
# test_gpu_ops.py are collected twice - 'fp32' (exact fp32 MFMA) and 'fp32x3' (three-plane tensors, six bf16 MFMAs per
# product block, csrc/conv_x3.hip) - with the SAME fp32 tolerances.  A module opts in by defining
#     pytest_generate_tests = conftest.both_math_modes(names or None)
#     math_mode = conftest.math_mode_fixture()
MATH_MODES = ['fp32', 'fp32x3']


def both_math_modes(names=None):
    def hook(metafunc):
        if 'math_mode' in metafunc.fixturenames and (names is None or metafunc.function.__name__ in names):
            metafunc.parametrize('math_mode', MATH_MODES, indirect=True)
    return hook


def math_mode_fixture(io_f32=True):
    @pytest.fixture(autouse=True)
    def math_mode(request):
        """'fp32x3': every 'fp32' request of the test (and the library default) runs as 'fp32x3'.  io_f32 (op-level test
        modules): direct calls of the conv / norm ops hand back fp32 copies of their three-plane results (ops.io_f32: the
        kernels still run on three-plane operands and results); model-level modules leave it off - the networks run exactly
        as in production, three-plane tensors from layer to layer."""
        mode = getattr(request, 'param', 'fp32')
        from iprgan import _lib, ops
        if mode == 'fp32x3':
            params = getattr(getattr(request.node, 'callspec', None), 'params', {})
            if any(str(v).startswith('bf16') for v in params.values()) or 'bf16' in request.node.name:
                pytest.skip('a bf16-mode case: one math mode of its own')
            _lib._FP32_VIA_X3 = True
            _lib.set_math('fp32')
            ops.io_f32(io_f32)
            assert _lib.get_math() == 'fp32x3'
        try:
            yield mode
        finally:
            if mode == 'fp32x3':
                _lib._FP32_VIA_X3 = False
                ops.io_f32(False)
                _lib.call('iprgan_debug_force_tiles', -1, -1)
                _lib.set_math('fp32')
    return math_mode
