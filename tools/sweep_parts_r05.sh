# three / four batch parts on per-part graphs with different stream picks (hardware-queue mapping diagnostics)
mkdir -p gpurun_out/r05/sw3
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-roofline"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 > gpurun_out/r05/sw3/$name.json; }
run s2 LDC_SPLIT=2
for m in -0 -1 -2 -3 0- 1- 00 12 23 30; do
  run s3_m$m LDC_SPLIT=3 LDC_PART_GRAPHS=2 LDC_AUX_FROM_SIDE=$m
done
run s2_m0 LDC_SPLIT=2 LDC_AUX_FROM_SIDE=0
run s2_m1 LDC_SPLIT=2 LDC_AUX_FROM_SIDE=1
run s2_m2 LDC_SPLIT=2 LDC_AUX_FROM_SIDE=2
run s2_m3 LDC_SPLIT=2 LDC_AUX_FROM_SIDE=3
