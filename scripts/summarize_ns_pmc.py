"""profiles/<out>_northstar_conv_pmc.json from the counter passes of scripts/conv_bench.py NS
(3x3 256->256 @64x64, batch 64, zero- and reflect-padded), collected as
    rocprofv3 --kernel-trace --pmc <set i> -d gpurun_out/prof -o ns<i> --output-format csv -- python3 scripts/conv_bench.py NS
with the sets  1: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU
               2: SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY
               3: SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_ANY
usage: python scripts/summarize_ns_pmc.py gpurun_out/prof profiles/r01"""
import collections
import csv
import json
import re
import sys


def main():
    prof, out = sys.argv[1:3]
    tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for i in (1, 2, 3):
        for r in csv.DictReader(open(f'{prof}/ns{i}_counter_collection.csv')):
            k = r['Kernel_Name']
            if 'gconv_kernel' in k or 'wgrad_kernel' in k or 'wgrad_t_kernel' in k:
                k = re.sub(r'\(.*', '', k).replace('void iprgan::', '')
                e = tot[k][r['Counter_Name']]
                e[0] += 1
                e[1] += float(r['Counter_Value'])
    res = {'_how': __doc__.split('usage')[0].strip() +
           ' Values are per-launch averages over all launches of the kernel (autotune trials included). '
           'mfma_util_pct = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs) * 100; '
           'lds_busy_pct = SQ_LDS_IDX_ACTIVE / 256 CUs / (GRBM_GUI_ACTIVE/8) * 100; '
           'valu_per_mfma = (SQ_INSTS_VALU - SQ_INSTS_MFMA) / SQ_INSTS_MFMA.', 'kernels': {}}
    for k, c in tot.items():
        v = {n: e[1] / max(e[0], 1) for n, e in c.items()}
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        res['kernels'][k] = {
            'launches': c['GRBM_GUI_ACTIVE'][0],
            'mfma_util_pct': round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024) * 100, 1),
            'lds_busy_pct': round(v['SQ_LDS_IDX_ACTIVE'] / 256 / cyc * 100, 1),
            'lds_bank_conflict_cycles': v['SQ_LDS_BANK_CONFLICT'],
            'lds_insts_per_mfma': round(v['SQ_INSTS_LDS'] / v['SQ_INSTS_MFMA'], 3),
            'valu_per_mfma': round((v['SQ_INSTS_VALU'] - v['SQ_INSTS_MFMA']) / v['SQ_INSTS_MFMA'], 3),
            'salu_per_mfma': round(v['SQ_INSTS_SALU'] / v['SQ_INSTS_MFMA'], 3),
            'wait_any_over_wave_cycles': round(v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], 3),
            'kernel_cycles': round(cyc),
        }
    json.dump(res, open(f'{out}_northstar_conv_pmc.json', 'w'), indent=1)
    print(json.dumps(res['kernels'], indent=1))


if __name__ == '__main__':
    main()
