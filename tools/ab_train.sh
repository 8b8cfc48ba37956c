#!/bin/bash
# round 6: same-box A/B of the c4 (optimisation step) line with the weight-gradient GEMMs on the side stream / in line
for rep in 1 2 3; do for OPT in "" "--no-dw-side"; do
  python bench.py --config c4 --steps 6 --warmup 2 --no-cpu-baseline $OPT 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$OPT]', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
