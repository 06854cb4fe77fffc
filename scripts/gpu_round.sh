#!/bin/bash
# One GPU-box pass that produces everything profiles/ holds for a round.  Run through gpurun from the repo root:
#   gpurun --timeout 2700 -- 'bash scripts/gpu_round.sh r04'
# then, back in the build container:
#   for w in "" srgan_ cyclegan_; do python scripts/summarize_profiles.py gpurun_out/prof <tag>_${w%_} profiles/<tag>_${w%_}; done
# (scripts/collect_round.sh does that).  rocprofv3 rules of this pool: the program goes directly after "--", counters
# are collected in their own passes (never together with a trace domain other than --kernel-trace).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O/prof
cd $R
if [ "$2" != "noprof-tests" ] && [ "$2" != "reprof" ]; then
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
fi
prof_workload () {   # $1 = workload, $2 = file tag, $3 = math mode, $4.. = step counts of the kernel-stats run
  local W=$1 T=$2 MATH=$3; shift 3
  # one un-profiled pass records the autotuner's choices; the profiler runs replay them (IPRGAN_TUNE_CACHE), so their
  # per-kernel averages contain the launches of the training step only, like the bench line's own HIP-event figures
  export IPRGAN_TUNE_CACHE=$O/tune_cache_$T.txt
  rm -f $IPRGAN_TUNE_CACHE
  cd $R
  timeout 600 python bench.py --workload $W --math $MATH --alt-math none --no-cpu-baseline "$@" > /dev/null 2> $O/tune_pass_$T.err
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o $T --output-format csv -- python3 $R/bench.py --workload $W --math $MATH --alt-math none --no-cpu-baseline "$@" > $O/${T}_bench_under_rocprof.json 2> $O/prof_$T.err
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof -o ${T}_fetch --output-format csv -- python3 $R/bench.py --workload $W --math $MATH --alt-math none --no-cpu-baseline --steps 4 --warmup 4 > /dev/null 2> $O/prof_${T}_fetch.err
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof -o ${T}_write --output-format csv -- python3 $R/bench.py --workload $W --math $MATH --alt-math none --no-cpu-baseline --steps 4 --warmup 4 > /dev/null 2> $O/prof_${T}_write.err
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/prof -o ${T}_mfma --output-format csv -- python3 $R/bench.py --workload $W --math $MATH --alt-math none --no-cpu-baseline --steps 4 --warmup 4 > /dev/null 2> $O/prof_${T}_mfma.err
  cd $R
  unset IPRGAN_TUNE_CACHE
}
if [ "$2" = "reprof" ]; then      # only the profiler passes of the listed file tags ("" = headline): $3 = "headline srgan ..."
  for w in $3; do
    case $w in
      headline) prof_workload dcgan64 ${TAG}_dcgan64_fp32x3 fp32x3 --steps 20 --warmup 8; t=${TAG}_dcgan64_fp32x3;;
      fp32) prof_workload dcgan64 ${TAG} fp32 --steps 20 --warmup 8; t=$TAG;;
      srgan) prof_workload srgan ${TAG}_srgan_fp32x3 fp32x3 --steps 8 --warmup 4; t=${TAG}_srgan_fp32x3;;
      cyclegan) prof_workload cyclegan ${TAG}_cyclegan_fp32x3 fp32x3 --steps 4 --warmup 2; t=${TAG}_cyclegan_fp32x3;;
    esac
    python scripts/summarize_profiles.py $O/prof $t $O/$t > /dev/null 2>> $O/summarize.err
  done
  find $O/prof -name '*kernel_trace.csv' -delete; find $O/prof -name '*counter_collection.csv' -delete; find $O/prof -name '*agent_info.csv' -delete
  ls $O | grep -c .; exit 0
fi
# headline math mode = fp32x3 (three-plane tensors); the exact-fp32 MFMA runs of the same workloads next to it
prof_workload dcgan64 ${TAG}_dcgan64_fp32x3 fp32x3 --steps 20 --warmup 8
prof_workload dcgan64 ${TAG} fp32 --steps 20 --warmup 8
prof_workload srgan ${TAG}_srgan_fp32x3 fp32x3 --steps 8 --warmup 4
prof_workload cyclegan ${TAG}_cyclegan_fp32x3 fp32x3 --steps 4 --warmup 2
prof_workload dcgan128 ${TAG}_dcgan128_bf16act bf16act --steps 8 --warmup 4
# the counters of THIS build first, so that the bench line's roofline.traffic (read from profiles/) matches it
for t in ${TAG}_dcgan64_fp32x3 ${TAG} ${TAG}_srgan_fp32x3 ${TAG}_cyclegan_fp32x3 ${TAG}_dcgan128_bf16act; do
  python scripts/summarize_profiles.py $O/prof $t $R/profiles/$t > /dev/null
done
cp $R/profiles/${TAG}*_pmc_traffic.json $R/profiles/${TAG}*_mfma_util.json $R/profiles/${TAG}*_bench_kernel_stats.csv $O/ 2>/dev/null
timeout 900 python bench.py > $O/${TAG}_bench.json 2> $O/bench.err; cut -c1-400 $O/${TAG}_bench.json
timeout 900 python bench.py --math fp32 --alt-math none --no-cpu-baseline > $O/${TAG}_bench_dcgan64_fp32.json 2>> $O/bench.err
for w in srgan cyclegan dcgan128; do
  timeout 900 python bench.py --workload $w > $O/${TAG}_bench_$w.json 2> $O/bench_$w.err; cut -c1-200 $O/${TAG}_bench_$w.json
  timeout 900 python bench.py --workload $w --math fp32 --alt-math none --no-cpu-baseline > $O/${TAG}_bench_${w}_fp32.json 2>> $O/bench_$w.err
done
timeout 900 python bench.py --workload dcgan128 --math bf16act --no-cpu-baseline > $O/${TAG}_bench_dcgan128_bf16act.json 2>> $O/bench_dcgan128.err
timeout 900 python bench.py --math bf16act --no-cpu-baseline > $O/${TAG}_bench_dcgan64_bf16act.json 2>> $O/bench.err
timeout 600 python scripts/conv_bench.py > $O/${TAG}_conv_bench.jsonl 2> $O/conv_bench.err
# three-plane tiles per layer (forward / backward-data by forced tile, backward-weight by candidate)
X3P_TILES=-1,18,19,21,26,28,32,33,34,36,37 timeout 900 python scripts/x3p_check.py bench > $O/${TAG}_conv_bench_fp32x3.jsonl 2>> $O/conv_bench.err
timeout 600 python scripts/probe/wgrad_x3_bench.py > $O/${TAG}_wgrad_bench_fp32x3.jsonl 2>> $O/conv_bench.err
# per-layer tables (conv-family launches by pass + geometry) of the workloads, headline math mode and exact fp32
for w in dcgan64 srgan cyclegan; do
  IPRGAN_BENCH_LAYERS=1 timeout 600 python bench.py --workload $w --alt-math none --no-cpu-baseline 2>&1 >/dev/null | grep -A200 "conv-family layers" | cut -c18- > $O/${TAG}_layers_${w}_fp32x3.txt
  IPRGAN_BENCH_LAYERS=1 timeout 600 python bench.py --workload $w --math fp32 --alt-math none --no-cpu-baseline 2>&1 >/dev/null | grep -A200 "conv-family layers" | cut -c18- > $O/${TAG}_layers_$w.txt
done
IPRGAN_BENCH_LAYERS=1 timeout 600 python bench.py --workload dcgan128 --math bf16act --no-cpu-baseline 2>&1 >/dev/null | grep -A200 "conv-family layers" | cut -c18- > $O/${TAG}_layers_dcgan128_bf16act.txt
# (rounds 4-5 also ran two / eight ranks on this box's ONE GPU over the test double tests/stub_rccl.cpp here: plumbing evidence
# that says nothing about scaling - VERDICT r05 asked not to spend GPU minutes on it again; tests/test_gpu_ddp.py keeps the coverage)
# one rank, the buckets through the real RCCL communicator of the C ABI (fork / ncclAllReduce / join inside the captured step)
IPRGAN_FORCE_COMM=1 timeout 600 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --alt-math none 2> $O/bench_1rank.err | grep '^{' > $O/${TAG}_bench_1rank_rccl.json      # (RCCL prints its version banner to stdout)
# the three-plane ring tile on the north-star shape: clock, MFMA-busy, LDS conflicts (scripts/probe/x3p_pmc.sh)
bash scripts/probe/x3p_pmc.sh 18 > /dev/null 2>&1; cp $O/pmc_x3p18.txt $O/${TAG}_northstar_x3p_pmc.txt 2>/dev/null
# ... and the clocks / power rocm-smi reports while that tile runs back to back (the power-bound reading as an observation)
bash scripts/probe/smi_log.sh $O/${TAG}_northstar_x3p_smi.txt > /dev/null 2>&1
# D.conv1 (64 -> 64 k4 s2 @64x64, batch 128) through the 256x64 tile: bytes from beyond L2 and L2 hit rate by K-walk order
# (0 = taps inside a chunk, row-major; 3 = the default for stride-2 gathers since round 5: parity-grouped taps, chunks innermost)
for K in 0 3; do IPRGAN_X3P_KORDER=$K bash scripts/probe/x3p_tcc.sh 22 2 tcck$K > /dev/null 2>&1; done
( echo "# IPRGAN_X3P_KORDER=0"; cat $O/pmc_tcck022_2.txt; echo "# IPRGAN_X3P_KORDER=3 (default for stride-2 gathers)"; cat $O/pmc_tcck322_2.txt ) > $O/${TAG}_dconv1_korder_tcc.txt 2>/dev/null
# north-star conv shape (3x3 256->256 @64x64, batch 64): counter passes for the per-kernel MFMA / LDS / VALU picture
export IPRGAN_TUNE_CACHE=$O/tune_cache_ns.txt
rm -f $IPRGAN_TUNE_CACHE
timeout 300 python scripts/conv_bench.py "256->256 k3" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU -d $O/prof -o ns1 --output-format csv -- python3 $R/scripts/conv_bench.py "256->256 k3" > /dev/null 2> $O/prof_ns1.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY -d $O/prof -o ns2 --output-format csv -- python3 $R/scripts/conv_bench.py "256->256 k3" > /dev/null 2> $O/prof_ns2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $O/prof -o ns3 --output-format csv -- python3 $R/scripts/conv_bench.py "256->256 k3" > /dev/null 2> $O/prof_ns3.err
cd $R
unset IPRGAN_TUNE_CACHE
python scripts/summarize_ns_pmc.py $O/prof $O/${TAG} > /dev/null 2> $O/ns_pmc.err
# ordered kernel list of one step per workload (what the launch diet of round 6 worked from)
for w in dcgan64 srgan; do STEPS=4 WARM=4 bash scripts/probe/step_trace.sh $w > /dev/null 2>&1; cp $O/step_trace_$w.txt $O/${TAG}_step_trace_$w.txt 2>/dev/null; done
# Everything judged is summarised HERE (gpurun merges at most 64 MiB back): per-workload kernel stats, HBM traffic and
# MFMA-busy summaries into $O, then the raw per-dispatch traces and counter dumps are dropped.
for t in ${TAG}_dcgan64_fp32x3 $TAG ${TAG}_srgan_fp32x3 ${TAG}_cyclegan_fp32x3 ${TAG}_dcgan128_bf16act; do
  python scripts/summarize_profiles.py $O/prof $t $O/$t > /dev/null 2>> $O/summarize.err
done
find $O/prof -name '*kernel_trace.csv' -delete
find $O/prof -name '*counter_collection.csv' -delete
find $O/prof -name '*agent_info.csv' -delete
du -sh $O; ls $O | head -80
