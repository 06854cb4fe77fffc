#!/bin/bash
for bk in 32 64 128; do
  for w in dcgan128 dcgan64; do
    IPRGAN_BF16_BK=$bk python bench.py --workload $w --math bf16 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w bk$bk', d['ms_per_step'], d['conv_kernels']['device_ms_per_step'], d['conv_kernels']['tflops'], [(k['name'], k['tflops']) for k in d['conv_kernels']['by_kernel']])"
  done
done
