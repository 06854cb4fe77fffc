#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -3 $O/pytest_gpu.log
export IPRGAN_TUNE_CACHE=/tmp/tune.bin
python bench.py --no-cpu-baseline --alt-math none > /dev/null 2>&1
for i in 1 2; do
for P in 0 1; do
  echo "PRIO=$P"; IPRGAN_X3WS_PRIO=$P python bench.py --no-cpu-baseline --alt-math none 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline']['achieved'])"
done; done
