#!/bin/bash
# L2 / fabric counters of one layer of scripts/x3p_check.py's BENCH list through one forced three-plane tile (gpurun):
#   bash scripts/probe/x3p_tcc.sh <tile> <layer index> [tag]      (environment, e.g. IPRGAN_X3P_KORDER, is inherited)
# FETCH_SIZE (x 2 on gfx950: MI355X_MICROARCH.md, HBM) = bytes that came from beyond the XCD's L2; TCC_HIT / TCC_MISS = L2 hit rate.
T=${1:-22}
LI=${2:-2}
TAG=${3:-tcc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O/pmc
cat > /tmp/x3p_one.py <<PY
import os, sys
sys.path.insert(0, '$R/scripts'); sys.path.insert(0, '$R/ipr-gan_amd')
os.environ['X3P_TILES'] = '$T'
import x3p_check
x3p_check.BENCH = x3p_check.BENCH[$LI:$LI + 1]
x3p_check.bench()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $O/pmc -o ${TAG}${T}_${LI}_$i --output-format csv -- python3 /tmp/x3p_one.py > /dev/null 2> $O/pmc/${TAG}${T}_${LI}_$i.err
done
cd $R
python3 scripts/pmc_probe_summary.py $O/pmc ${TAG}${T}_${LI} > $O/pmc_${TAG}${T}_$LI.txt
find $O/pmc -name '*.csv' -delete
cat $O/pmc_${TAG}${T}_$LI.txt
