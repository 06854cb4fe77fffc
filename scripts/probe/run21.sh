#!/bin/bash
for NS2 in 0 1; do echo "== tile 36 NS2=$NS2"; IPRGAN_X3WS_NS2=$NS2 timeout 300 python scripts/probe/tile_overhead.py 36 2>&1 | tail -2
IPRGAN_X3WS_NS2=$NS2 X3P_TILES=36 timeout 600 python scripts/x3p_check.py bench 2>&1 | tail -12 | cut -c1-120; done
