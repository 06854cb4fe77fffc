#!/bin/bash
# Quick kernel-stats profile of a bench.py invocation (tuning replayed from a cache so that the stats hold the step's
# launches only).  usage: gpurun -- 'bash scripts/gpu_prof.sh <tag> [bench args...]'
TAG=${1:-prof}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O/prof
export IPRGAN_TUNE_CACHE=$O/tune_cache_$TAG.txt
rm -f $IPRGAN_TUNE_CACHE
cd $R
timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline "$@" > /dev/null 2> $O/tune_pass.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o $TAG --output-format csv -- python3 $R/bench.py --steps 20 --warmup 8 --no-cpu-baseline "$@" > $O/${TAG}_bench_under_rocprof.json 2> $O/prof_bench.err
cd $R
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
python - <<PY
import csv, glob
f = sorted(glob.glob('$O/prof/**/${TAG}_kernel_stats.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total device ms', tot / 1e6)
for r in rows[:28]:
    print(f"{r['Name'][:70]:70s} n={int(r['Calls']):6d} tot={float(r['TotalDurationNs'])/1e6:8.2f}ms avg={float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
PY
cut -c1-300 $O/${TAG}_bench_under_rocprof.json
