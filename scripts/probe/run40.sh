#!/bin/bash
O=gpurun_out; mkdir -p $O
for i in 1 2 3 4 5 6; do timeout 900 python -m pytest tests/test_gpu_models.py -m gpu -q -k "late_step" -p no:cacheprovider 2>&1 | grep -E "passed|failed|AssertionError:" | cut -c1-300; done
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -E "^FAILED|passed|failed" $O/pytest_gpu.log | tail -4
