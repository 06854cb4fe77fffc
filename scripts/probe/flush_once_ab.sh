# A/B of the captured one-rank flush policy (parallel.py: _FLUSH_ONCE) on the headline step + the tests that walk the executor
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
IPRGAN_FLUSH_AT_BUCKETS=1 python bench.py --steps 100 --warmup 20 --alt-math none --no-cpu-baseline --north-star off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flush at buckets', j['ms_per_step'], j.get('eager_ms_per_step'), j.get('host_enqueue_ms_per_eager_step'))"
python bench.py --steps 100 --warmup 20 --alt-math none --no-cpu-baseline --north-star off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flush once      ', j['ms_per_step'], j.get('eager_ms_per_step'), j.get('host_enqueue_ms_per_eager_step'))"
done
python -m pytest tests/test_gpu_models.py tests/test_gpu_driver.py tests/test_gpu_ddp.py -q -m gpu -p no:cacheprovider > gpurun_out/flush_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/flush_tests.log | tail -3
