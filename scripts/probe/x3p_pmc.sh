#!/bin/bash
# Counter passes over one layer of scripts/x3p_check.py's BENCH list (default 0 = the north-star 3x3 layer) through one forced
# three-plane tile (gpurun):  bash scripts/probe/x3p_pmc.sh 18 [layer index]
T=${1:-18}
LI=${2:-0}
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
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
           "SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $O/pmc -o x3p${T}_${LI}_$i --output-format csv -- python3 /tmp/x3p_one.py > /dev/null 2> $O/pmc/x3p${T}_${LI}_$i.err
done
cd $R
python3 scripts/pmc_probe_summary.py $O/pmc x3p${T}_${LI} > $O/pmc_x3p${T}_$LI.txt; [ $LI = 0 ] && cp $O/pmc_x3p${T}_$LI.txt $O/pmc_x3p$T.txt
find $O/pmc -name '*.csv' -delete
cat $O/pmc_x3p${T}_$LI.txt
