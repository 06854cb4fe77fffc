#!/bin/bash
# Per-layer conv-family table of a workload's step (IPRGAN_BENCH_LAYERS): usage  bash scripts/dbg/layers.sh <workload> [bench args]
W=${1:-dcgan64}; shift
IPRGAN_BENCH_LAYERS=1 timeout 600 python bench.py --workload $W --no-cpu-baseline "$@" 2>&1 >/dev/null | grep -A200 "conv-family layers" | cut -c1-170
