#!/bin/bash
# round 6: same-box A/B of the quick c2 line: conv_fast_kernel everywhere (LDC_OPTIONS=conv_lean=0) vs the default, interleaved
for rep in 1 2 3; do for LEAN in 0 1; do
  LDC_OPTIONS=conv_lean=$LEAN python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lean=$LEAN', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
