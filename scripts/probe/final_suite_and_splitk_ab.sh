cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/pytest_gpu_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_final.log
tail -3 gpurun_out/pytest_gpu_final.log
for i in 1 2; do
python bench.py --workload srgan --steps 30 --warmup 10 --north-star off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', j['ms_per_step'])"
IPRGAN_SPLITK_BLOCKS=2400 IPRGAN_SPLITK_OUT=10000000 python bench.py --workload srgan --steps 30 --warmup 10 --north-star off 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('splitk-wide', j['ms_per_step'])"
done
