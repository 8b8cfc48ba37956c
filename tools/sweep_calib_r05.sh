# stream-overlap calibration: decode time with 0..4 foreign streams created in front of the context's part streams, calibration on / off
mkdir -p gpurun_out/r05/cal
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-roofline"
run() { name=$1; shift; env "$@" $B 2>gpurun_out/r05/cal/$name.err | tail -1 > gpurun_out/r05/cal/$name.json; }
for n in 0 1 2 3 4; do
  run x${n}_cal LDC_TEST_EXTRA_STREAMS=$n LDC_VERBOSE=1
  run x${n}_nocal LDC_TEST_EXTRA_STREAMS=$n LDC_NO_STREAM_CALIB=1
done
run s3_cal LDC_SPLIT=3 LDC_PART_GRAPHS=2
run s4_cal LDC_SPLIT=4 LDC_PART_GRAPHS=2
grep -h "part streams" gpurun_out/r05/cal/*.err | sort | uniq -c
