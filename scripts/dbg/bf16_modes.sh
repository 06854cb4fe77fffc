#!/bin/bash
for m in bf16 bf16act; do
  for w in dcgan128 dcgan64; do
    python bench.py --workload $w --math $m --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$w $m', d['value'], d['ms_per_step'], d['conv_kernels']['device_ms_per_step'], d['conv_kernels']['tflops'], [(k['name'], k['launches'], k['tflops']) for k in d['conv_kernels']['by_kernel']], d['metrics_last_step'])"
  done
done
