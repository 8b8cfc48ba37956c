#!/bin/bash
# same-box A/B of two builds of the library (LDC_LIB_PATH):  bash tools/ab_lib.sh ladiffcodec_amd/lib_old.so ladiffcodec_amd/libladiffcodec.so
for rep in 1 2 3; do for LIB in "$@"; do
  LDC_LIB_PATH="$PWD/$LIB" python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$LIB]', round(d['value'],1), round(d['ms_per_step'],2))"
done; done
