#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q -x 2>&1 | tail -5
P=$PWD/ipr-gan_amd/iprgan/libiprgan_prev.so
for L in prev new prev new; do
  if [ $L = prev ]; then export IPRGAN_LIB=$P; else unset IPRGAN_LIB; fi
  echo "== $L"; timeout 300 python scripts/probe/tile_overhead.py 36 2>&1 | tail -2 | head -1
  timeout 300 python scripts/probe/tile_overhead.py 35 2>&1 | tail -2 | head -1
  timeout 300 python scripts/probe/tile_overhead.py 37 2>&1 | tail -2 | head -1
done
for L in prev new; do
  if [ $L = prev ]; then export IPRGAN_LIB=$P; else unset IPRGAN_LIB; fi
  echo "== $L"; X3P_TILES=35,36,37 timeout 600 python scripts/x3p_check.py bench 2>&1 | tail -12 | cut -c1-200
done
export IPRGAN_TUNE_CACHE=/tmp/tune.bin
for L in prev new prev new; do
  if [ $L = prev ]; then export IPRGAN_LIB=$P; else unset IPRGAN_LIB; fi
  echo "== bench $L"; python bench.py --no-cpu-baseline --alt-math none 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print(j['ms_per_step'], j['roofline']['kernel'], j['roofline']['frac'], j['roofline'].get('ring_family',{}).get('frac'))"
done
