# tile / split-K thresholds in the two-batches-in-flight mode (every launch carries 32 items)
run() { echo -n "$1: "; shift; env "$@" python bench.py --in-flight 2 --steps 6 --no-cpu-baseline --no-roofline --no-pipelined 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), round(r['ms_per_step'],2))"; }
for rep in 1 2; do
run base LDC_X=1
run sk320 LDC_SK_TILES=320
run sk400 LDC_SK_TILES=400
run small120 LDC_CONV_SMALL_TILES=120
run small160 LDC_CONV_SMALL_TILES=160
run small30 LDC_CONV_SMALL_TILES=30
run nosplitk LDC_CONV_SPLITK=0
run graph10 LDC_GRAPH_STEPS=10
run graph2 LDC_GRAPH_STEPS=2
run notail LDC_NO_TAIL_FUSE=1
done
