cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -q -x -k "deferred or conv_fwd_bwd or every_gconv_tile or splitk or north_star" > gpurun_out/t1.log 2>&1; tail -3 gpurun_out/t1.log
python -m pytest tests/test_gpu_models.py -q -x -k "deferred or dcgan_steps or bs128 or srgan_steps or cyclegan_steps or late_step" > gpurun_out/t2.log 2>&1; tail -3 gpurun_out/t2.log
python -m pytest tests/test_gpu_x3.py -q -x > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log
IPRGAN_BENCH_LAYERS=1 python bench.py --no-cpu-baseline --alt-math none > gpurun_out/bench2.json 2> gpurun_out/bench2.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/bench2.json'))
print(r['value'], r['ms_per_step'], r['ms_per_step_median'], r['roofline']['frac'], r['roofline']['achieved'])
for k in r['conv_kernels']['by_kernel']: print(k)
PY
grep -A60 "conv-family layers" gpurun_out/bench2.err | cut -c18-150
