"""CPU: the C-ABI library loads (no GPU needed) and exports every symbol include/iprgan.h declares;
the ctypes table in iprgan/_lib.py covers exactly the same set."""
import os
import re

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'iprgan.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return set(re.findall(r'\b(iprgan_\w+)\s*\(', text))


def test_header_matches_ctypes_table():
    from iprgan import _lib
    assert header_symbols() == set(_lib.SIGNATURES), (
        sorted(header_symbols() ^ set(_lib.SIGNATURES)))


def test_library_exports_every_symbol():
    from iprgan import _lib
    lib = _lib.load()                      # raises if the .so is missing: there is no fallback
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.iprgan_version() == int(re.search(r'#define IPRGAN_VERSION (\d+)', open(os.path.join(ROOT, 'include', 'iprgan.h')).read()).group(1))
    assert lib.iprgan_last_error() is not None


def test_size_queries_run_without_gpu():
    import ctypes as C
    from iprgan import _lib, ops
    spec = ops.ConvSpec(256, 512, 3, 1, 1)
    d = spec.desc(128, 8, 8)
    assert _lib.query('iprgan_conv_wfwd_floats', C.byref(d)) == 512 * 9 * 256
    assert _lib.query('iprgan_conv_wbwd_floats', C.byref(d)) == 256 * 9 * 512
    assert _lib.query('iprgan_conv_wgrad_ws_floats', C.byref(d)) > 512 * 9 * 256
    d3 = ops.ConvSpec(3, 64, 3, 1, 1).desc(2, 16, 16)
    assert _lib.query('iprgan_conv_wfwd_floats', C.byref(d3)) == 128 * 64     # K = 9 taps x 4 ch -> 64


def test_tune_table_round_trips_without_gpu():
    """iprgan_tune_export / iprgan_tune_import (host-only): what parallel.sync_autotune ships from rank 0 to its peers."""
    from iprgan import parallel
    before = parallel.tune_table()
    rec = [[7, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 21], [7, 9, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 18]]
    parallel.tune_adopt(rec, replace=False)
    after = parallel.tune_table()
    assert all(r in after for r in rec) and len(after) == len(before) + 2
    parallel.tune_adopt([[7, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 30]], replace=False)      # overwrite by key
    assert [r[16] for r in parallel.tune_table() if r[:2] == [7, 1]] == [30]
    parallel.tune_adopt(before, replace=True)
    assert parallel.tune_table() == before
