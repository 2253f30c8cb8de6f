#!/bin/bash
# C3 / C4 with and without the index clamp of the cell lookup (desc.cell_noclamp), and C5 both streams.
set -e
for c in c3 c4; do
  echo "== $c default"; python bench.py --config $c --no-cpu-baseline --no-cold --steps 30 --warmup 5
  echo "== $c clamped"; MCX_NO_NOCLAMP=1 python bench.py --config $c --no-cpu-baseline --no-cold --steps 30 --warmup 5
done
echo "== c5"; python bench.py --config c5 --no-cpu-baseline --no-cold --steps 10 --warmup 3
