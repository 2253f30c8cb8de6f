#!/bin/bash
# physical threads per launch (the logical grid is re-cut into this many), every config
for c in c2 c3 c4 c5; do
  S=40; [ $c = c4 ] && S=10; [ $c = c5 ] && S=8
  for t in 262144 524288 786432 1048576 1572864 2097152 4194304; do
    python bench.py --config $c --no-cpu-baseline --no-cold --no-philox --steps $S --warmup 4 --target-phys $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c target_phys=$t', '%.4g' % d['value'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
  done
done
