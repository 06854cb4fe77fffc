#!/bin/bash
# Ordered kernel list (name, duration) of ONE step of a workload under rocprofv3 --kernel-trace: what the launch diet works from.
# gpurun: bash scripts/probe/step_trace.sh [workload] [math]   -> gpurun_out/step_trace_<workload>.txt
W=${1:-dcgan64}; M=${2:-fp32x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O/prof
export IPRGAN_TUNE_CACHE=$O/tune_cache_trace_$W.txt
cd $R && timeout 600 python bench.py --workload $W --math $M --alt-math none --no-cpu-baseline --north-star off --steps ${STEPS:-6} --warmup ${WARM:-6} > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $O/prof -o trace_$W --output-format csv -- python3 $R/bench.py --workload $W --math $M --alt-math none --no-cpu-baseline --north-star off --graph off --steps ${STEPS:-6} --warmup ${WARM:-6} > /dev/null 2> $O/trace_$W.err
cd $R
python3 - "$O/prof" "trace_$W" "$O/step_trace_$W.txt" <<'PY'
import csv, glob, sys, re
d, tag, out = sys.argv[1:4]
f = sorted(glob.glob(f'{d}/**/{tag}_kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# one step = between two consecutive sign_loss_fwd launches (exactly one per step in every workload: the white-box term of
# update_g), taken from the tail of the run
idx = [i for i, n in enumerate(names) if 'sign_loss_fwd' in n]
a, b = idx[-3], idx[-2]
with open(out, 'w') as o:
    tot = 0
    for r in rows[a:b]:
        us = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        tot += us
        n = re.sub(r'^void ', '', r['Kernel_Name']).replace('iprgan::', '')
        o.write(f'{us:9.1f} us  {n[:150]}\n')
    span = (int(rows[b - 1]['End_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3
    o.write(f'# {b - a} launches, kernel time {tot:.1f} us, span {span:.1f} us\n')
print(open(out).read()[-300:])
PY
find $O/prof -name '*kernel_trace.csv' -delete; find $O/prof -name '*agent_info.csv' -delete
