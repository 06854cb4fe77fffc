"""Per-kernel averages of every counter collected by scripts/pmc_probe.sh (conv-family kernels only)."""
import collections
import csv
import glob
import re
import sys


def main():
    prof, tag = sys.argv[1:3]
    tot = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob(f'{prof}/{tag}_*_counter_collection.csv')):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if not re.search(r"gconv|wgrad_halo|fewin", k):
                continue
            k = re.sub(r'\(.*', '', k).replace('void iprgan::', '')
            e = tot[k][r['Counter_Name']]
            e[0] += 1
            e[1] += float(r['Counter_Value'])
    for f in sorted(glob.glob(f'{prof}/{tag}_1_kernel_trace.csv')):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void iprgan::', '')
            if k in tot:
                dur[k][0] += 1
                dur[k][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for k, c in tot.items():
        print(k, f'launches={max(e[0] for e in c.values())}', f'avg_us={dur[k][1] / max(dur[k][0], 1):.1f}')
        for n, e in sorted(c.items()):
            print(f'    {n:32s} {e[1] / max(e[0], 1):16.1f}')


if __name__ == '__main__':
    main()
