#!/bin/bash
# Register / scratch / occupancy summary of every kernel in one HIP source (device pass only):
#   scripts/kernel_resources.sh ipr-gan_amd/csrc/conv_pipe.hip
F=$1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$(dirname $0)/../include --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage -c $F -o /dev/null 2>&1 | python3 -c '
import sys, re, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: rows.append(cur)
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    else:
        cur[k.split()[0]] = v
if cur: rows.append(cur)
for r in rows:
    print("%-110s vgpr %4s agpr %4s scratch %5s occ %s" % (r["name"][:110], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
'
