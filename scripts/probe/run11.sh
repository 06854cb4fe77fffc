cd $GRAFT_REPO_ROOT
timeout 600 python scripts/x3p_check.py acc > gpurun_out/acc.jsonl 2>&1; tail -1 gpurun_out/acc.jsonl; grep -c '"ok": false' gpurun_out/acc.jsonl
X3P_TILES=-1,22,32,19,33,18 timeout 900 python scripts/x3p_check.py bench > gpurun_out/wsbench.jsonl 2> gpurun_out/wsbench.err; cat gpurun_out/wsbench.jsonl
