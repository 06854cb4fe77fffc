#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round.  Run through gpurun from the repo root:
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round.sh r01'
# then, back in the build container:
#   python scripts/summarize_profiles.py gpurun_out/prof <tag> profiles/<tag>
# rocprofv3 rules of this pool: the program goes directly after "--", counters are collected in their own passes.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O/prof
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
# one un-profiled pass records the autotuner's choices; the profiler runs replay them (IPRGAN_TUNE_CACHE), so their
# per-kernel averages contain the launches of the training step only, like the bench line's own HIP-event figures
export IPRGAN_TUNE_CACHE=$O/tune_cache.txt
rm -f $IPRGAN_TUNE_CACHE
timeout 300 python bench.py --steps 8 --warmup 4 --no-cpu-baseline > /dev/null 2> $O/tune_pass.err
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o $TAG --output-format csv -- python3 $R/bench.py --steps 20 --warmup 8 --no-cpu-baseline > $O/${TAG}_bench_under_rocprof.json 2> $O/prof_bench.err
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof -o ${TAG}_fetch --output-format csv -- python3 $R/bench.py --steps 4 --warmup 8 --no-cpu-baseline > /dev/null 2> $O/prof_fetch.err
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof -o ${TAG}_write --output-format csv -- python3 $R/bench.py --steps 4 --warmup 8 --no-cpu-baseline > /dev/null 2> $O/prof_write.err
cd /tmp
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof -o ${TAG}_mfma --output-format csv -- python3 $R/bench.py --steps 4 --warmup 8 --no-cpu-baseline > /dev/null 2> $O/prof_mfma.err
cd $R
# the counters of THIS build first, so that the bench line's roofline.traffic (read from profiles/) matches it
python scripts/summarize_profiles.py $O/prof $TAG $R/profiles/$TAG > /dev/null && cp $R/profiles/${TAG}_pmc_traffic.json $R/profiles/${TAG}_bench_kernel_stats.csv $O/
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err; cut -c1-400 $O/${TAG}_bench.json
cd $R
timeout 600 python scripts/conv_bench.py > $O/${TAG}_conv_bench.jsonl 2> $O/conv_bench.err
timeout 900 python scripts/step_bench.py > $O/${TAG}_step_bench.jsonl 2> $O/step_bench.err
# the raw per-dispatch traces are large; only the stats and counter CSVs are needed back
find $O/prof -name '*kernel_trace.csv' -size +20M -delete
ls -la $O/prof | head -20
