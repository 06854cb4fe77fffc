import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import bench
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = bench.build_model(dev)
x = torch.tanh(torch.randn(128, 3, 64, 64, device=dev)); z = torch.randn(128, 128, device=dev)
for i in range(15): bench.step(m, x, z)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for i in range(40): bench.step(m, x, z)
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f'wall {t/40*1e3:.3f} ms/step, host-loop {th/40*1e3:.3f}', flush=True)
