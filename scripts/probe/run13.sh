cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for E in 0 1; do
  for w in dcgan64 srgan cyclegan; do
    IPRGAN_X3WS=$E python bench.py --workload $w --no-cpu-baseline --alt-math none > gpurun_out/ab_${w}_$E.json 2>/dev/null
    python -c "import json; r=json.load(open('gpurun_out/ab_${w}_$E.json')); print('$w ws=$E', r['ms_per_step'], r['ms_per_step_median'], r['roofline']['frac'])"
  done
done
done
