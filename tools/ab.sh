# A/B two builds of the library on one box: ladiffcodec_amd/libprev.so vs libladiffcodec.so
for rep in 1 2 3; do
  for which in prev new; do
    if [ $which = prev ]; then export LDC_LIB_PATH=$PWD/ladiffcodec_amd/libprev.so; else unset LDC_LIB_PATH; fi
    echo -n "$which: "; timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value'],1), round(r['ms_per_step'],2))"
  done
done
