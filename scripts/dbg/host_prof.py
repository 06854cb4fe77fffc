"""cProfile of the host side of a workload's step (which Python functions the enqueueing thread spends its time in).
usage: python scripts/dbg/host_prof.py <workload> [steps]"""
import cProfile, gc, io, os, pstats, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import bench
from iprgan import Config, models, _lib
_lib.set_math(os.environ.get('HOST_PROF_MATH', 'fp32'))
name = sys.argv[1] if len(sys.argv) > 1 else 'srgan'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
torch.manual_seed(0)
model, step = bench.make_workload(name, [dev], (Config, models))
for i in range(6):
    step(i)
torch.cuda.synchronize()
gc.collect(); gc.freeze()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
th = time.perf_counter() - t0
torch.cuda.synchronize()
t = time.perf_counter() - t0
print(f'{name}: wall {t / steps * 1e3:.3f} ms/step, host loop {th / steps * 1e3:.3f} ms/step')
torch.autograd.set_multithreading_enabled(False)      # backward on this thread: visible to cProfile
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
pr.disable()
torch.cuda.synchronize()
for key in ('tottime', 'cumulative'):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:60]))
