#!/bin/bash
# where the host's time goes when it replays the step graphs: chains x graph arrangement (bench JSON `host` object)
cd ${GRAFT_REPO_ROOT:-$PWD}
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-pipelined"
run() { name=$1; shift; echo "$name: $(env "$@" timeout 300 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['host'])" 2>&1 | tail -1)"; }
for rep in 1 2; do
run forkjoin_graph LDC_PART_GRAPHS=0
run part_graphs LDC_PART_GRAPHS=1
done
run part_graphs_k10 LDC_PART_GRAPHS=1 LDC_GRAPH_STEPS=10
run part_graphs_k25 LDC_PART_GRAPHS=1 LDC_GRAPH_STEPS=25
run part_graphs_k2 LDC_PART_GRAPHS=1 LDC_GRAPH_STEPS=2
run part_graphs_split3 LDC_PART_GRAPHS=1 LDC_SPLIT=3
run part_graphs_split4 LDC_PART_GRAPHS=1 LDC_SPLIT=4
run split1 LDC_SPLIT=1
