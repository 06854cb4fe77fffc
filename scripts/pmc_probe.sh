#!/bin/bash
# Counter passes over one layer of the bf16 microbench (run on the GPU box through gpurun):
#   bash scripts/pmc_probe.sh "D.conv1" "12"          # layer filter, forced tile list (CONV_BENCH_TILES)
# Each pass is its own rocprofv3 run with --kernel-trace only (pool rule); the summary goes to gpurun_out/pmc_<tag>.txt
L=$1; T=$2; TAG=${3:-probe}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O/pmc
export CONV_BENCH_TILES=$T
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d $O/pmc -o ${TAG}_$i --output-format csv -- python3 $R/scripts/conv_bench_bf16.py "$L" > /dev/null 2> $O/pmc/${TAG}_$i.err
done
cd $R
python3 scripts/pmc_probe_summary.py $O/pmc $TAG > $O/pmc_$TAG.txt
find $O/pmc -name '*.csv' -delete
cat $O/pmc_$TAG.txt
