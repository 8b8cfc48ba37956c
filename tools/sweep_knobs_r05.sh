mkdir -p gpurun_out/r05/sw5
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-roofline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 > gpurun_out/r05/sw5/$name.json; }
run base A=1
run k10 LDC_GRAPH_STEPS=10
run k25 LDC_GRAPH_STEPS=25
run k2 LDC_GRAPH_STEPS=2
run d2 LDC_FLOW_DEPTH=2
run d1 LDC_FLOW_DEPTH=1
run nap0 LDC_GN_NAP=0
run nap2 LDC_GN_NAP=2
run nap4 LDC_GN_NAP=4
run nap00 LDC_GN_NAP0=0
run nap08 LDC_GN_NAP0=8
run base2 A=1
