"""cProfile of the HOST side of eager DCGAN-64 steps (what an N > 1 run without capture pays per step): the model is warmed up
(autotune, operand caches), then ``N`` steps are enqueued under the profiler with the device left to run behind (no sync inside
the window), so `tottime` is host work, not waiting.  gpurun: python scripts/probe/host_profile.py [workload] [steps]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch  # noqa: E402
import bench  # noqa: E402
from iprgan import Config, _lib, models  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else 'dcgan64'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
_lib.set_math('fp32x3')
torch.manual_seed(1)
model, step = bench.make_workload(wl, [torch.device('cuda:0')], (Config, models))
for i in range(12):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(N):
    step(i)
host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f'{wl}: host enqueue {host / N * 1e3:.3f} ms/step (unprofiled), wall {(time.perf_counter() - t0) / N * 1e3:.3f} ms/step')
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step(i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(30)
st.sort_stats('cumulative').print_stats(30)
# the backward passes run on the autograd engine's own (C++-created) thread, which the profiler above does not see: profile the
# bodies of the executor's backward functions from inside that thread
from iprgan import engine, tools  # noqa: E402
prb = cProfile.Profile()
orig = engine.ChainFn.backward


def wrapped(ctx, dy):
    prb.enable()
    try:
        return orig(ctx, dy)
    finally:
        prb.disable()


engine.ChainFn.backward = staticmethod(wrapped)
for i in range(N):
    step(i)
torch.cuda.synchronize()
print('==== inside ChainFn.backward (autograd thread), %d steps' % N)
sb = pstats.Stats(prb)
sb.sort_stats('tottime').print_stats(35)
sb.sort_stats('cumulative').print_stats(30)
