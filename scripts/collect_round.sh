#!/bin/bash
# After `gpurun -- 'bash scripts/gpu_round.sh <tag>'`: copy the judged summaries from gpurun_out/ into profiles/.
TAG=${1:-r06}
cd "$(dirname "$0")/.."
for t in ${TAG}_dcgan64_fp32x3 $TAG ${TAG}_srgan_fp32x3 ${TAG}_cyclegan_fp32x3 ${TAG}_dcgan128_bf16act; do     # summarised on the GPU box by gpu_round.sh
  for sfx in bench_kernel_stats.csv pmc_traffic.json mfma_util.json bench_under_rocprof.json; do
    cp gpurun_out/${t}_$sfx profiles/ 2>/dev/null
  done
done
for f in gpurun_out/${TAG}_bench*.json gpurun_out/${TAG}_conv_bench*.jsonl gpurun_out/${TAG}_wgrad_bench*.jsonl \
         gpurun_out/${TAG}_northstar_*.json gpurun_out/${TAG}_northstar_*.txt gpurun_out/${TAG}_layers_*.txt gpurun_out/${TAG}_dconv1_*.txt \
         gpurun_out/${TAG}_step_trace_*.txt; do
  [ -s "$f" ] && cp "$f" profiles/
done
tail -3 gpurun_out/pytest_gpu.log > profiles/${TAG}_pytest_gpu_tail.txt 2>/dev/null
ls -la profiles | grep $TAG
