cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x -k "split_tiles and (35 or 36 or 37)" > gpurun_out/t1.log 2>&1; tail -2 gpurun_out/t1.log
X3P_TILES=-1,32,35,33,36,34,37,28 timeout 900 python scripts/x3p_check.py bench > gpurun_out/wsbench3.jsonl 2> gpurun_out/wsbench3.err
for E in 0 1; do IPRGAN_X3WS=$E python bench.py --no-cpu-baseline --alt-math none > gpurun_out/ab_$E.json 2>/dev/null; python -c "import json; r=json.load(open('gpurun_out/ab_$E.json')); print('dcgan ws=$E', r['ms_per_step'], r['ms_per_step_median'], r['roofline']['frac'])"; done
