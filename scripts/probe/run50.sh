#!/bin/bash
O=gpurun_out; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_rep$i.log 2>&1; echo "run $i rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest_gpu_rep$i.log | tail -4
done
cp $O/pytest_gpu_rep2.log $O/pytest_gpu.log
python bench.py 2>/dev/null | tail -1 | cut -c1-400
