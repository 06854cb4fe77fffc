#!/bin/bash
# After `gpurun -- 'bash scripts/gpu_round.sh <tag>'`: copy the judged summaries from gpurun_out/ into profiles/.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
for t in $TAG ${TAG}_srgan ${TAG}_cyclegan ${TAG}_dcgan128_bf16act ${TAG}_dcgan64_fp32x3; do     # summarised on the GPU box by gpu_round.sh
  for sfx in bench_kernel_stats.csv pmc_traffic.json mfma_util.json bench_under_rocprof.json; do
    cp gpurun_out/${t}_$sfx profiles/ 2>/dev/null
  done
done
for f in ${TAG}_bench.json ${TAG}_bench_srgan.json ${TAG}_bench_cyclegan.json ${TAG}_bench_dcgan128.json \
         ${TAG}_bench_dcgan128_bf16.json ${TAG}_bench_dcgan64_bf16.json ${TAG}_bench_dcgan128_bf16act.json \
         ${TAG}_bench_dcgan64_bf16act.json ${TAG}_bench_dcgan64_fp32x3.json ${TAG}_conv_bench_fp32x3.jsonl ${TAG}_conv_bench.jsonl ${TAG}_conv_bench_bf16.jsonl ${TAG}_northstar_conv_pmc.json ${TAG}_layers_dcgan64.txt \
         ${TAG}_layers_srgan.txt ${TAG}_layers_cyclegan.txt ${TAG}_layers_dcgan128_bf16act.txt; do
  cp gpurun_out/$f profiles/ 2>/dev/null
done
ls -la profiles | grep $TAG
