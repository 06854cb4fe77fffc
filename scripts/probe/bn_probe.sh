#!/bin/bash
# gpurun: per-kernel averages of the norm kernels on the three BatchNorm tensors of ConvGenerator64 (batch 128), fp32 and three-plane
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O/bn
cd /tmp && export TMPDIR=/tmp
for cfg in "128 8 256" "128 16 128" "128 32 64" "64 24 64"; do
  for kind in 0 2; do
    t=$(echo $cfg | tr ' ' _)_$kind
    timeout 200 rocprofv3 --kernel-trace --stats -d $O/bn -o $t --output-format csv -- python3 $R/scripts/probe/bn_probe.py $cfg $kind > /dev/null 2> $O/bn/$t.err
    echo "== $cfg kind $kind"; grep -E "bn_|colreduce|colsum" $O/bn/${t}_kernel_stats.csv | awk -F, '{printf "%-60s calls=%s avg_us=%.1f\n", substr($1,1,60), $2, $4/1000}'
  done
done > $O/bn_probe.txt 2>&1
find $O/bn -name '*.csv' -delete
cat $O/bn_probe.txt
