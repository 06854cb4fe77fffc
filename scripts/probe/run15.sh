cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "(test_pipe_tiles or test_bf16_gconv_tiles) and (35 or 36 or 37)" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
for E in 0 1; do IPRGAN_PIPE_WS=$E IPRGAN_BENCH_LAYERS=1 python bench.py --workload dcgan128 --math bf16act --no-cpu-baseline --alt-math none > gpurun_out/c5_$E.json 2> gpurun_out/c5_$E.err; python -c "import json; r=json.load(open('gpurun_out/c5_$E.json')); print('config5 ws=$E', r['ms_per_step'], r['ms_per_step_median'], r['roofline'])"; done
grep -A26 "conv-family layers" gpurun_out/c5_1.err | cut -c18-150
for E in 0 1; do IPRGAN_PIPE_WS=$E python bench.py --math bf16act --no-cpu-baseline --alt-math none > gpurun_out/c2b_$E.json 2>/dev/null; python -c "import json; r=json.load(open('gpurun_out/c2b_$E.json')); print('dcgan64 bf16act ws=$E', r['ms_per_step'])"; done
