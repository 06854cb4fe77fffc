cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_x3.py -q -k "planes or exponent or cancellation or non_finite" > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
python -m pytest tests/test_gpu_models.py -q -s -k late_step > gpurun_out/t2.log 2>&1; grep -E "moment tensors|passed|failed|Error" gpurun_out/t2.log | cut -c1-400
python scripts/probe/final_moment_dist.py > gpurun_out/moments.log 2>&1; grep -E "^\[" gpurun_out/moments.log
for K in 0 1; do IPRGAN_X3P_KORDER=$K X3P_TILES=-1,18,20,21,22 timeout 600 python scripts/x3p_check.py bench > gpurun_out/kbench_$K.jsonl 2> gpurun_out/kbench_$K.err; done
cat gpurun_out/kbench_0.jsonl gpurun_out/kbench_1.jsonl
IPRGAN_X3P_KORDER=0 bash scripts/probe/x3p_tcc.sh 22 2 tcck0 > /dev/null 2>&1
IPRGAN_X3P_KORDER=1 bash scripts/probe/x3p_tcc.sh 22 2 tcck1 > /dev/null 2>&1
cat gpurun_out/pmc_tcck0*.txt gpurun_out/pmc_tcck1*.txt
