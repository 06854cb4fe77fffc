"""Per-kernel totals of a step trace written by scripts/probe/step_trace.sh: python scripts/probe/trace_summary.py <file> [top]"""
import collections
import re
import sys

agg = collections.OrderedDict()
tot = 0.0
for ln in open(sys.argv[1]):
    if ln.startswith('#'):
        print(ln.strip())
        continue
    m = re.match(r'\s*([\d.]+) us\s+(.*)', ln)
    us, n = float(m.group(1)), m.group(2).split('(')[0][:90]
    a = agg.setdefault(n, [0, 0.0])
    a[0] += 1
    a[1] += us
    tot += us
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f'{us:9.1f} us {100 * us / tot:5.1f} % {c:4d}  {n}')
