#!/bin/bash
# A/B of the few-input-channel direct convolution (IPRGAN_FEWIN) on the RGB layers and on the steps they appear in.
O=gpurun_out; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "conv" 2>&1 | tail -2
for f in 1 0; do
  echo "== FEWIN=$f"
  IPRGAN_FEWIN=$f timeout 300 python scripts/conv_bench.py 2>/dev/null | grep -E "D.conv0|G.out" | cut -c1-260
  IPRGAN_FEWIN=$f timeout 300 python bench.py --workload dcgan128 --math bf16act --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | cut -c1-200
  IPRGAN_FEWIN=$f timeout 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | cut -c1-200
done
