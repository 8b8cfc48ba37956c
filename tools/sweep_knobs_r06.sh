#!/bin/bash
# round 6: one-box sweep of the launch-structure knobs with the lean conv kernel (quick c2 line, 6 steps)
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', round(d['value'],1), round(d['ms_per_step'],2))"; }
run base A=1
run split3 LDC_OPTIONS=split=3,part_graphs=2
run split4 LDC_OPTIONS=split=4,part_graphs=2
run small30 LDC_OPTIONS=conv_small_tiles=30
run small100 LDC_OPTIONS=conv_small_tiles=100
run sk300 LDC_OPTIONS=sk_tiles=300
run sk100 LDC_OPTIONS=sk_tiles=100
run nosk LDC_OPTIONS=conv_splitk=0
run nap4 LDC_OPTIONS=gn_nap=4
run nap32 LDC_OPTIONS=gn_nap=32
run k10 LDC_OPTIONS=graph_steps=10
run base2 A=1
