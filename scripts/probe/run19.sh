#!/bin/bash
for T in 36 37; do
for P in 0 256; do
  echo "== tile $T probe $P"; IPRGAN_X3WS_PROBE=$P python scripts/probe/tile_overhead.py $T 2>&1 | grep -v "^$" | tail -16
done; done
echo "== tile 29 (x3p16 128x128, no loader waves)"; python scripts/probe/tile_overhead.py 29 2>&1 | tail -3
