#!/bin/bash
for i in 1 2; do timeout 300 python bench.py --math bf16act --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"; done
IPRGAN_FEWIN=0 timeout 300 python bench.py --math bf16act --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fewin0', d['ms_per_step'], d.get('host_enqueue_ms_per_step'))"
IPRGAN_BENCH_LAYERS=1 timeout 300 python bench.py --math bf16act --no-cpu-baseline 2>&1 >/dev/null | grep -A45 "conv-family" | cut -c1-160
