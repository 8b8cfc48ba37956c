#!/bin/bash
# round 6: same-box A/B of the quick c2 line over LDC_OPTIONS settings, interleaved:  bash tools/ab_opt.sh "" "fold_ctx=0" ...
for rep in 1 2 3; do for OPT in "$@"; do
  LDC_OPTIONS="$OPT" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$OPT]', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
