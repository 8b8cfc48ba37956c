#!/bin/bash
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"; }
run base A=1
for sm in 80 100 150 200 300 1000; do run small$sm LDC_OPTIONS=conv_small_tiles=$sm; run small${sm}_nosk LDC_OPTIONS=conv_small_tiles=$sm,conv_splitk=0; done
run sk_u2_48 LDC_OPTIONS=sk_u2=48
run base2 A=1
