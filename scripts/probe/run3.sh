cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_models.py -q -s -k "late_step or steps_vs_reference_golden or complete_protection" > gpurun_out/t2.log 2>&1; grep -E "moment tensors|passed|failed|Error" gpurun_out/t2.log | cut -c1-300
for K in 2 3; do IPRGAN_X3P_KORDER=$K X3P_TILES=-1,18,21,22 timeout 600 python scripts/x3p_check.py bench > gpurun_out/kbench_$K.jsonl 2> gpurun_out/kbench_$K.err; done
cat gpurun_out/kbench_2.jsonl gpurun_out/kbench_3.jsonl
for S in 1500 3000 5000; do IPRGAN_X3P_KORDER=1 IPRGAN_X3P_STAGGER_NS=$S X3P_TILES=-1,18 timeout 600 python scripts/x3p_check.py bench > gpurun_out/sbench_$S.jsonl 2> gpurun_out/sbench_$S.err; echo "stagger $S"; cat gpurun_out/sbench_$S.jsonl; done
IPRGAN_X3P_KORDER=2 bash scripts/probe/x3p_tcc.sh 22 2 tcck2 > /dev/null 2>&1
IPRGAN_X3P_KORDER=3 bash scripts/probe/x3p_tcc.sh 22 2 tcck3 > /dev/null 2>&1
cat gpurun_out/pmc_tcck2*.txt gpurun_out/pmc_tcck3*.txt
