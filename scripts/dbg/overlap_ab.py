"""dgrad and wgrad of one layer: back to back on one stream vs on two streams (fork/join with events)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'ipr-gan_amd'))
import torch
from iprgan import ops
LAYERS = [  # name, B, cin, cout, k, s, p, transposed, H, reflect
    ('srgan res 64->64 k3 @24', 64, 64, 64, 3, 1, 1, False, 24, 0),
    ('srgan vgg 512->512 @6', 64, 512, 512, 3, 1, 1, False, 6, 0),
    ('srgan vgg 512->512 @12', 64, 512, 512, 3, 1, 1, False, 12, 0),
    ('srgan D 512->512 s2 @12', 64, 512, 512, 3, 2, 1, False, 12, 0),
    ('srgan vgg 256->256 @24', 64, 256, 256, 3, 1, 1, False, 24, 0),
    ('dcgan D.conv2 64->128 k3 @32', 128, 64, 128, 3, 1, 1, False, 32, 0),
    ('dcgan D.conv1 64->64 k4s2 @64', 128, 64, 64, 4, 2, 1, False, 64, 0),
    ('dcgan G.up1 256->128 T @16', 128, 256, 128, 4, 2, 1, True, 16, 0),
    ('cyclegan res 256->256 reflect @64', 8, 256, 256, 3, 1, 1, False, 64, 1),
]
dev = torch.device('cuda:0')
side = torch.cuda.Stream()


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for name, b, cin, cout, k, s, p, tr, H, refl in LAYERS:
    spec = ops.ConvSpec(cin, cout, k, s, p, 0, tr, pad_mode=refl)
    d = spec.desc(b, H, H)
    OH, OW = spec.out_hw(H, H)
    x = torch.randn(b, H, H, ops.c4(cin), device=dev)
    dy = torch.randn(b, OH, OW, ops.c4(cout), device=dev)
    wshape = (cin, cout, k, k) if tr else (cout, cin, k, k)
    w = torch.randn(*wshape, device=dev) * 0.05
    wf, wb = ops.conv_prep(spec, d, w, None, True, True)
    ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()

    def serial():
        ops.conv_bwd_data(spec, d, dy, wb)
        ops.conv_bwd_weight(spec, d, x, dy, wshape, False)

    def forked():
        main = torch.cuda.current_stream()
        ev_f.record(main)
        side.wait_event(ev_f)
        with torch.cuda.stream(side):
            ops.conv_bwd_weight(spec, d, x, dy, wshape, False)
            ev_j.record(side)
        ops.conv_bwd_data(spec, d, dy, wb)
        main.wait_event(ev_j)

    t_d = timeit(lambda: ops.conv_bwd_data(spec, d, dy, wb))
    t_w = timeit(lambda: ops.conv_bwd_weight(spec, d, x, dy, wshape, False))
    t_s = timeit(serial)
    t_f = timeit(forked)
    print(json.dumps(dict(layer=name, dgrad_us=round(t_d, 1), wgrad_us=round(t_w, 1), serial_us=round(t_s, 1),
                          forked_us=round(t_f, 1))), flush=True)
