cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_x3.py -q -x -k "persistent" > gpurun_out/t1.log 2>&1; tail -12 gpurun_out/t1.log | cut -c1-300
X3P_TILES=-1,19,33,21,34,23,35 timeout 900 python scripts/x3p_check.py bench > gpurun_out/pbench.jsonl 2> gpurun_out/pbench.err; cat gpurun_out/pbench.jsonl
