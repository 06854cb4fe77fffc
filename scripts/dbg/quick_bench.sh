#!/bin/bash
# one bench line per workload / math mode, compact
show () { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], 'host', d.get('host_enqueue_ms_per_step'), 'roof', d.get('step_roofline_frac'), d['roofline']['frac'])"; }
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | show dcgan64
timeout 300 python bench.py --no-cpu-baseline --math bf16act 2>/dev/null | show dcgan64-bf16act
timeout 300 python bench.py --no-cpu-baseline --workload srgan 2>/dev/null | show srgan
timeout 300 python bench.py --no-cpu-baseline --workload cyclegan 2>/dev/null | show cyclegan
timeout 300 python bench.py --no-cpu-baseline --workload dcgan128 --math bf16act 2>/dev/null | show dcgan128-bf16act
