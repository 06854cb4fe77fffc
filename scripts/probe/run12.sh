cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_x3.py -q -x -k "split_tiles" > gpurun_out/t1.log 2>&1; tail -2 gpurun_out/t1.log
IPRGAN_BENCH_LAYERS=1 python bench.py --no-cpu-baseline --alt-math none > gpurun_out/bench3.json 2> gpurun_out/bench3.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/bench3.json'))
print(r['value'], r['ms_per_step'], r['ms_per_step_median'], r['roofline']['frac'], r['roofline']['achieved'])
for k in r['conv_kernels']['by_kernel']: print(k)
PY
grep -A48 "conv-family layers" gpurun_out/bench3.err | cut -c18-150
for w in srgan cyclegan; do python bench.py --workload $w --no-cpu-baseline --alt-math none > gpurun_out/ws_$w.json 2>/dev/null; python -c "import json; r=json.load(open('gpurun_out/ws_$w.json')); print('$w', r['ms_per_step'], r['roofline']['frac'])"; done
