"""Prints the DESIGN.md section 5 table rows from profiles/<tag>_bench*.json (one gpu_round.sh run).
usage: python scripts/design_table.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'


def load(name):
    p = os.path.join(ROOT, 'profiles', f'{tag}_{name}.json')
    if not os.path.exists(p):
        return None
    return json.loads(open(p).read().strip().splitlines()[-1])


rows = [('DCGAN-64 B=128 fp32 (headline)', 'bench'), ('SRGAN 24→96 B=64', 'bench_srgan'),
        ('CycleGAN Resnet9 256² B=8', 'bench_cyclegan'), ('DCGAN-128 B=256 fp32', 'bench_dcgan128'),
        ('DCGAN-128 B=256 `bf16`', 'bench_dcgan128_bf16'), ('DCGAN-128 B=256 `bf16act`', 'bench_dcgan128_bf16act'),
        ('DCGAN-64 B=128 `bf16`', 'bench_dcgan64_bf16'), ('DCGAN-64 B=128 `bf16act`', 'bench_dcgan64_bf16act')]
for label, f in rows:
    j = load(f)
    if j is None:
        continue
    ck, rf, cb = j.get('conv_kernels', {}), j.get('roofline') or {}, j.get('cpu_baseline') or {}
    g = j.get('graph') or {}
    print(f"| {label} | **{j['ms_per_step']:.2f}** | {j['value']:.1f} {j['unit']} | {100 * j['step_roofline_frac']:.1f} % | "
          f"{ck.get('device_ms_per_step')} ms at {ck.get('tflops')} TFLOP/s; dominant `{rf.get('kernel')}` {rf.get('frac')} of peak "
          f"({rf.get('achieved')} TFLOP/s, traffic {rf.get('traffic')}) | {ck.get('mfma_util_pct_pmc')} % | "
          f"{cb.get('value', '')} {cb.get('unit', '')} ({cb.get('cores', '')} threads) | host {j.get('host_enqueue_ms_per_step')} ms, "
          f"graph replays {g.get('replays_in_timed_region')} of {j['steps']} |")
