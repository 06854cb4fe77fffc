#!/bin/bash
# bf16 / bf16act bench lines of the DCGAN workloads (+ per-layer table of DCGAN-128 bf16act)
python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "bf16" 2>&1 | tail -3
IPRGAN_TUNE_LOG=1 IPRGAN_BENCH_LAYERS=1 timeout 600 python bench.py --workload dcgan128 --math bf16act --no-cpu-baseline --steps 30 --warmup 8 2> gpurun_out/d128a.err | cut -c1-200
grep "tune\] gconv" gpurun_out/d128a.err | cut -c1-150
grep -A60 "conv-family layers" gpurun_out/d128a.err | cut -c1-170
timeout 600 python bench.py --workload dcgan128 --math bf16 --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | cut -c1-200
timeout 600 python bench.py --math bf16act --no-cpu-baseline 2>/dev/null | cut -c1-200
