"""Where a loader-wave tile spends its time: shader-clock stamps of multiplying wave 0 of every block (debug build of conv_x3.hip with
-DIPRGAN_X3WS_TIMING, loaded through IPRGAN_LIB).  Stamps: 0 entry, 1 first barrier passed (prologue landed), 2 K loop done, 3 final
barrier passed + accumulators added, 4 accumulators in LDS (barrier passed), 5 row passes issued, 6 stores acknowledged.
gpurun: IPRGAN_LIB=$PWD/ipr-gan_amd/iprgan/libiprgan_dbg.so python scripts/probe/ws_phase_times.py [tile]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from iprgan import ops, _lib  # noqa: E402

dev = torch.device('cuda:0')
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 36
_lib.set_math('fp32x3')
lib = _lib.load()
LAYERS = [('NS 256->256 k3 @64 B64', 256, 256, 3, 1, 1, False, 64, 64), ('D.conv4 128->256 k3 B256', 128, 256, 3, 1, 1, False, 16, 256),
          ('D.conv2 64->128 k3 B256', 64, 128, 3, 1, 1, False, 32, 256), ('D.conv1 64->64 k4s2 B256', 64, 64, 4, 2, 1, False, 64, 256)]
for name, cin, cout, k, s, p, tr, H, B in LAYERS:
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr)
    d = spec.desc(B, H, H)
    x = ops.to_kind(torch.randn(B, H, H, cin, device=dev), 2)
    w = torch.randn(cout, cin, k, k, device=dev) * 0.05
    wf, _ = ops.conv_prep(spec, d, w, None, True, False)
    _lib.call('iprgan_debug_force_tiles', tile, -1)
    for _ in range(3):
        ops.conv_fwd(spec, d, x, wf, None)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ops.conv_fwd(spec, d, x, wf, None); b.record()
    torch.cuda.synchronize()
    _lib.call('iprgan_debug_force_tiles', -1, -1)
    n = 8192 * 8
    buf = (C.c_ulonglong * n)()
    rc = lib.iprgan_debug_x3ws_ts(buf, C.c_size_t(n))
    ts = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 8).astype(np.int64)
    OH = spec.out_hw(H, H)[0]
    M = B * OH * OH
    bm, bn = {36: (128, 128), 35: (256, 64), 37: (128, 64)}[tile]
    nb = min(8192, (M // bm) * max(1, cout // bn))
    t = ts[:nb]
    t = t[(t[:, :7] > 0).all(axis=1)]          # (column 7: cycles of wave 0 inside the K-loop barrier, summed over the steps)
    dif = np.diff(t[:, :7], axis=1).astype(np.float64)          # shader-clock cycles (s_memtime; every XCD has its own base: only differences inside a block mean anything)
    names = ['prologue (entry -> stage 0 landed)', 'K loop', 'final barrier + acc sum', 'acc -> LDS + barrier', 'row passes', 'store ack']
    steps = cin * k * k // 32
    row = dict(layer=name, tile=tile, blocks=int(len(t)), launch_us=round(a.elapsed_time(b) * 1e3, 1), k_steps=steps,
               mean_cycles={n_: int(dif[:, i].mean()) for i, n_ in enumerate(names)},
               p90_cycles={n_: int(np.percentile(dif[:, i], 90)) for i, n_ in enumerate(names)},
               cycles_per_k_step=round(float(dif[:, 1].mean() / steps), 1), block_cycles=int((t[:, 6] - t[:, 0]).mean()),
               barrier_wait_cycles_per_k_step=round(float(t[:, 7].mean() / max(1, steps - 1)), 1))
    print(json.dumps(row), flush=True)
