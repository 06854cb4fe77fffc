cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_models.py -q -x -k "thread_local or communicator" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
bash scripts/gpu_round.sh r05 reprof headline > gpurun_out/reprof.log 2>&1
head -40 gpurun_out/r05_dcgan64_fp32x3_bench_kernel_stats.csv
for m in bf16act; do python bench.py --workload dcgan128 --math $m --no-cpu-baseline --alt-math none > gpurun_out/b128_$m.json 2> gpurun_out/b128_$m.err; cut -c1-600 gpurun_out/b128_$m.json; done
IPRGAN_PIPE_KORDER=49 python bench.py --workload dcgan128 --math bf16act --no-cpu-baseline --alt-math none > gpurun_out/b128_k49.json 2> gpurun_out/b128_k49.err; cut -c1-300 gpurun_out/b128_k49.json
