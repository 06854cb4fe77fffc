# config 5 (DCGAN-128 B=256 bf16act): paired D pass (possible since can_pair counts bf16 bytes) against two passes
cd $GRAFT_REPO_ROOT
for i in 1 2; do
IPRGAN_PAIR_D=0 python bench.py --workload dcgan128 --math bf16act --steps 30 --warmup 8 --alt-math none --no-cpu-baseline --north-star off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two passes', j['ms_per_step'], j.get('eager_ms_per_step'))"
python bench.py --workload dcgan128 --math bf16act --steps 30 --warmup 8 --alt-math none --no-cpu-baseline --north-star off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('paired    ', j['ms_per_step'], j.get('eager_ms_per_step'))"
done
python -m pytest tests/test_gpu_models.py -q -m gpu -p no:cacheprovider -k "128 or pair" > gpurun_out/pair128_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/pair128_tests.log | tail -3
