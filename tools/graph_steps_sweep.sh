run() { echo -n "$1: "; shift; env "$@" python bench.py --in-flight 2 --steps 6 --no-cpu-baseline --no-roofline --no-pipelined 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), round(r['ms_per_step'],2))"; }
for rep in 1 2; do
run g5 LDC_GRAPH_STEPS=5
run g8 LDC_GRAPH_STEPS=8
run g10 LDC_GRAPH_STEPS=10
run g13 LDC_GRAPH_STEPS=13
run g17 LDC_GRAPH_STEPS=17
run g25 LDC_GRAPH_STEPS=25
done
