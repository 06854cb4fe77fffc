#!/bin/bash
# gpurun: per-kernel averages of the norm kernels on the three BatchNorm tensors of ConvGenerator64 (batch 128) and SRResNet's
# (batch 64, 24x24), fp32 and three-plane
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O/bn
cd /tmp && export TMPDIR=/tmp
for cfg in "128 8 256" "128 16 128" "128 32 64" "64 24 64" "16 64 256"; do
  for kind in 0 2; do
    t=$(echo $cfg | tr ' ' _)_$kind
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/bn -o $t --output-format csv -- python3 $R/scripts/probe/bn_probe.py $cfg $kind > /dev/null 2> $O/bn/$t.err
    echo "== B H C = $cfg kind $kind"
    python3 - $O/bn/${t}_kernel_stats.csv <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r'\(.*', '', r['Name']).replace('void iprgan::', '').replace('iprgan::', '')
    if re.search('bn_|colreduce|colsum', n):
        print(f"   {n:44s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs']) / 1e3:7.1f}")
PY
  done
done > $O/bn_probe.txt 2>&1
find $O/bn -name '*.csv' -delete
cat $O/bn_probe.txt
