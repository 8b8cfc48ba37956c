mkdir -p gpurun_out/r05/tall
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "every_tile_shape" 2>&1 | tail -3
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-roofline"
run() { name=$1; shift; env "$@" $B 2>gpurun_out/r05/tall/$name.err | tail -1 > gpurun_out/r05/tall/$name.json; }
run base A=1
run force3 LDC_TILE_CFG=3
run t250 LDC_CONV_TALL_TILES=250
run t140 LDC_CONV_TALL_TILES=140
run t70 LDC_CONV_TALL_TILES=70
run base2 A=1
