#!/bin/bash
# flush period (units accumulated in f32 between folds into f64) x workgroup size, every config
for c in c2 c3 c4 c5; do
  S=40; [ $c = c4 ] && S=10; [ $c = c5 ] && S=8
  for b in 0 256 512 1024; do
    for f in 32 64 128 256; do
      [ $b = 0 ] && unset MCX_BLOCK || export MCX_BLOCK=$b
      MCX_EXTRA_DEFINES="MCX_FLUSH=$f" python bench.py --config $c --no-cpu-baseline --no-cold --no-philox --steps $S --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c block=$b flush=$f', '%.4g' % d['value'], round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4))"
    done
  done
done
