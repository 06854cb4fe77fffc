import sys, os, time, gc, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ipr-gan_amd')]
import bench
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = bench.build_model(dev)
xs = [torch.tanh(torch.randn(128, 3, 64, 64, device=dev)) for _ in range(8)]
zs = [torch.randn(128, 128, device=dev) for _ in range(8)]
if os.environ.get('NOGC'): gc.disable()
st = []
for i in range(60):
    if i in (15, 40):
        torch.cuda.synchronize()
        if os.environ.get('SLEEP'): time.sleep(float(os.environ['SLEEP']))
    t0 = time.perf_counter()
    bench.step(m, xs[i % 8], zs[i % 8])
    st.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print(' '.join(f'{t*1e3:.1f}' for t in st))
