mkdir -p gpurun_out/r05/skx
python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shape.py -m gpu -x -q 2>&1 | tail -2
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-roofline"
for i in 1 2 3; do
  $B 2>/dev/null | tail -1 > gpurun_out/r05/skx/on$i.json
  LDC_NO_SK_XCD=1 LDC_NO_XCD_GRID42=1 $B 2>/dev/null | tail -1 > gpurun_out/r05/skx/off$i.json
done
bash tools/run_r05.sh pmc > gpurun_out/r05/skx/pmc.log 2>&1; tail -12 gpurun_out/r05/skx/pmc.log
