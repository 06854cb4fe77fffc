cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q -x -k "reflect or instance_norm or batchnorm" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
python -m pytest tests/test_gpu_models.py -q -x -k "cyclegan or net_vs_reference or net_accuracy" > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
IPRGAN_BENCH_LAYERS=1 python bench.py --workload cyclegan --no-cpu-baseline --alt-math none > gpurun_out/bcyc.json 2> gpurun_out/bcyc.err; cut -c1-330 gpurun_out/bcyc.json
grep -A14 "conv-family layers" gpurun_out/bcyc.err | cut -c18-150
IPRGAN_REFLECT_DIRECT=0 python bench.py --workload cyclegan --no-cpu-baseline --alt-math none > gpurun_out/bcyc0.json 2> gpurun_out/bcyc0.err; cut -c1-330 gpurun_out/bcyc0.json
