# 128x64 vs 128x128 tiles by grid size (LDC_CONV_MEDIUM_TILES = largest grid, in 128x128-equivalents, that still uses 128x64)
run() { echo -n "$1: "; shift; env "$@" python bench.py --steps 6 --no-cpu-baseline --no-roofline --no-pipelined $EXTRA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), round(r['ms_per_step'],2))"; }
for mode in "--in-flight 2" ""; do
  EXTRA=$mode
  echo "== mode '$mode'"
  for rep in 1 2; do
    run default LDC_X=1
    run m600 LDC_CONV_MEDIUM_TILES=600
    run m300 LDC_CONV_MEDIUM_TILES=300
    run m150 LDC_CONV_MEDIUM_TILES=150
    run m0 LDC_CONV_MEDIUM_TILES=0
  done
done
