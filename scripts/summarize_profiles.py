"""Turn rocprofv3 CSVs (gpurun_out/prof/<tag>_*.csv) into the committed summaries under profiles/:
  <out>_bench_kernel_stats.csv   copy of <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats)
  <out>_pmc_traffic.json         HBM bytes per launch per kernel from the FETCH_SIZE / WRITE_SIZE passes
Kernel names are mapped to the names bench.py reports (iprgan_prof_get).  FETCH_SIZE is doubled as
MI355X_MICROARCH.md prescribes for gfx950 (128-B requests of 16-B/lane streams are tallied at 64 B).
usage: python scripts/summarize_profiles.py <prof_dir> <tag> <out_prefix>"""
import collections
import os
import csv
import json
import re
import shutil
import sys


def short(k):
    """rocprof kernel name -> the name bench.py reports (iprgan_prof_get slots)."""
    m = re.match(r'void iprgan::gconv_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (\d+), (\d+)(?:, (true|false))?(?:, (true|false))?(?:, (true|false))?(?:, (true|false))?>', k)
    if m:
        if m.group(11) == 'true':         # SPLIT: math mode fp32x3
            return 'gconv_x3_kernel'
        if m.group(8) == 'true':
            return 'gconv_bf16_kernel'
        wgm, wgn, wm, wn = [int(x) for x in m.groups()[:4]]
        return f'gconv_kernel<{wgm * wm * 32}x{wgn * wn * 32}' + (',8w>' if wgm * wgn == 8 else '>')
    m = re.match(r'void iprgan::gconv_pipe2?_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false)(?:, (true|false))?(?:, (true|false))?>', k)
    if m:           # LDS-DMA ring tiles (conv_pipe.hip): profiling slots 19 / 20 by tile width, 23 for the fp32 form
        if 'gconv_pipe_kernel' in k and m.group(8) == 'true':
            return 'gconv_pipe_f32_kernel'
        return 'gconv_pipe_kernel' if int(m.group(2)) * int(m.group(4)) * 32 >= 128 else 'gconv_pipe_kernel<256x64>'
    m = re.match(r'void iprgan::(wgrad_halo_f32_kernel|wgrad_halo_kernel|gconv_phase4_kernel|gconv_pipe8_kernel|gconv_x3h_kernel|wgrad_x3h_kernel)<', k)
    if m:
        return m.group(1)
    m = re.match(r'void iprgan::(gconv_x3p_kernel|gconv_x3p16_kernel|gconv_x3ws_kernel)<', k)      # three-plane ring tiles (conv_x3.hip):
    if m:                                                                                           # profiling slots 29 / 32 / 33
        return m.group(1)
    m = re.match(r'void iprgan::wgrad_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (true|false)(?:, (true|false))?(?:, (true|false))?(?:, (true|false))?>', k)
    if m:
        if m.group(9) == 'true':
            return 'wgrad_x3_kernel'
        if m.group(7) == 'true':
            return 'wgrad_bf16_kernel'
        wgm, wgn, wm, wn = [int(x) for x in m.groups()[:4]]
        return f'wgrad_kernel<{wgm * wm * 32}x{wgn * wn * 32}' + (',8w>' if wgm * wgn == 8 else '>')
    m = re.match(r'void iprgan::wgrad_t_kernel<(\d+), (\d+), (\d+), (\d+), (true|false), (true|false)>', k)
    if m:
        wgm, wgn, wm, wn = [int(x) for x in m.groups()[:4]]
        return f'wgrad_t_kernel<{wgm * wm * 32}x{wgn * wn * 32}' + (',8w>' if wgm * wgn == 8 else '>')
    return k.split('(')[0].replace('void ', '').replace('iprgan::', '')


def agg(path, cname):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == cname:
            d[short(r['Kernel_Name'])][0] += 1
            d[short(r['Kernel_Name'])][1] += float(r['Counter_Value'])
    return d


def main():
    prof, tag, out = sys.argv[1:4]
    shutil.copy(f'{prof}/{tag}_kernel_stats.csv', f'{out}_bench_kernel_stats.csv')
    f, w = agg(f'{prof}/{tag}_fetch_counter_collection.csv', 'FETCH_SIZE'), agg(f'{prof}/{tag}_write_counter_collection.csv', 'WRITE_SIZE')
    res = {'_how': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (with --kernel-trace only) over '
                   '`bench.py [--workload W] --steps 4 --warmup 8` (see scripts/gpu_round.sh) with the tile choices replayed from IPRGAN_TUNE_CACHE (no tuning launches); '
                   'counter unit KiB; FETCH_SIZE x2 (gfx950 correction, checked on bn_apply whose byte count is known); '
                   'averages over all launches of a kernel name (layers of different sizes share kernels).',
           'kernels': {}}
    for k in sorted(f):
        n, wn = f[k][0], max(1, w[k][0])
        fb, wb = 2 * f[k][1] / n * 1024, w[k][1] / wn * 1024
        res['kernels'][k] = {'launches_sampled': n, 'fetch_bytes_per_launch': round(fb), 'write_bytes_per_launch': round(wb),
                             'hbm_bytes_per_launch': round(fb + wb)}
    json.dump(res, open(f'{out}_pmc_traffic.json', 'w'), indent=1)
    print('wrote', f'{out}_pmc_traffic.json', len(res['kernels']), 'kernels')
    mf = f'{prof}/{tag}_mfma_counter_collection.csv'
    if os.path.exists(mf):          # hardware MFMA utilisation of the conv kernels of the DCGAN step itself
        busy, gui = agg(mf, 'SQ_VALU_MFMA_BUSY_CYCLES'), agg(mf, 'GRBM_GUI_ACTIVE')
        util = {'_how': 'rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `bench.py --steps 4 '
                        '--warmup 8` (tile choices replayed from IPRGAN_TUNE_CACHE); mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES '
                        '/ (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs) * 100, summed over all launches of a kernel name; '
                        '"all_conv" weighs every gconv / wgrad launch of the step by its cycles.', 'kernels': {}}
        tb = tg = 0.0
        for k in sorted(busy):
            if ('gconv' not in k and 'wgrad' not in k and 'fewin_mfma' not in k) or 'wgrad_reduce' in k:
                continue
            b, g = busy[k][1], gui[k][1]
            tb += b
            tg += g
            util['kernels'][k] = {'launches_sampled': busy[k][0], 'mfma_util_pct': round(b / (g / 8 * 1024) * 100, 1)}
        util['all_conv'] = {'mfma_util_pct': round(tb / (tg / 8 * 1024) * 100, 1)}
        json.dump(util, open(f'{out}_mfma_util.json', 'w'), indent=1)
        print('wrote', f'{out}_mfma_util.json', util['all_conv'])


if __name__ == '__main__':
    main()
