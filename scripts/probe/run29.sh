#!/bin/bash
O=gpurun_out; mkdir -p $O
for i in 1 2 3; do timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "late_step" -s 2>&1 | grep -E "passed|failed|moment tensors|Error" | cut -c1-400; done
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py --no-cpu-baseline --alt-math none --graph off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('eager', j['ms_per_step'], j.get('graph'))"
