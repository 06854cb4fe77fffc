import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import bench
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = bench.build_model(dev)
x = torch.tanh(torch.randn(128, 3, 64, 64, device=dev)); z = torch.randn(128, 128, device=dev)
out = []
for blk in range(16):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(10): bench.step(m, x, z)
    torch.cuda.synchronize()
    out.append(round((time.perf_counter() - t0) / 10 * 1e3, 2))
print('ms/step per block of 10:', out)
print('mem', torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20)
