#!/bin/bash
# Clocks and power of the GPU while the three-plane ring tile multiplies the north-star layer back to back (VERDICT r04 next
# #8c: the power-bound reading of profiles/*_northstar_x3p_pmc.txt as an observation).  gpurun: bash scripts/probe/smi_log.sh <out>
OUT=${1:-gpurun_out/northstar_x3p_smi.txt}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cat > /tmp/x3p_loop.py <<PY
import os, sys, time
sys.path.insert(0, '$R/scripts'); sys.path.insert(0, '$R/ipr-gan_amd')
import torch
from iprgan import ops, _lib
_lib.set_math('fp32x3')
dev = torch.device('cuda:0')
spec = ops.ConvSpec(256, 256, 3, 1, 1, 0, False)
d = spec.desc(64, 64, 64)
x = ops.to_kind(torch.randn(64, 64, 64, 256, device=dev), 2)
w = torch.randn(256, 256, 3, 3, device=dev) * 0.05
wf, _ = ops.conv_prep(spec, d, w, None, True, False)
_lib.call('iprgan_debug_force_tiles', 18, -1)
for _ in range(5): ops.conv_fwd(spec, d, x, wf, None)
torch.cuda.synchronize()
t0 = time.time(); n = 0
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
while time.time() - t0 < float(sys.argv[1]):
    for _ in range(50): ops.conv_fwd(spec, d, x, wf, None)
    torch.cuda.synchronize(); n += 50
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f'north-star forward through tile 18, {n} launches back to back: {ms * 1e3:.1f} us = {309.24 / ms:.1f} TFLOP/s fp32-equivalent')
PY
echo "# idle" > $OUT
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|fclk" >> $OUT
python3 /tmp/x3p_loop.py 12 > /tmp/x3p_loop.out 2>&1 &
PID=$!
sleep 5
for i in 1 2 3 4 5; do
  echo "# under load, sample $i" >> $OUT
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|fclk" >> $OUT
  sleep 1
done
wait $PID
cat /tmp/x3p_loop.out >> $OUT
cat $OUT
