timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python scripts/conv_bench.py > gpurun_out/ab_n.jsonl 2> gpurun_out/ab_n.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/abb_n.json 2> gpurun_out/abb_n.err
