#!/bin/bash
# cell address out of the mantissa (desc.cell_addr16) against v_cvt_u32_f32 (MCX_NO_ADDR16=1): C3, C4
for c in c3 c4; do S=30; [ $c = c4 ] && S=10
  for v in 0 1 0 1; do
    if [ $v = 1 ]; then export MCX_NO_ADDR16=1; else unset MCX_NO_ADDR16; fi
    python bench.py --config $c --no-cpu-baseline --no-cold --no-philox --steps $S --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c no_addr16=$v', '%.4g' % d['value'], round(d['ms_per_step'],4))"
  done
done
