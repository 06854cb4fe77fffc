#!/bin/bash
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3.py -m gpu -q -x -k "persistent or 38 or 39" 2>&1 | tail -15
for T in 38 39; do echo "== tile $T"; timeout 300 python scripts/probe/tile_overhead.py $T 2>&1 | tail -16; done
X3P_TILES=-1,36,37,38,39 timeout 600 python scripts/x3p_check.py bench 2>&1 | tail -14
