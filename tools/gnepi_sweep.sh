#!/bin/bash
# round 6: stamps of the fused GroupNorm epilogue in the conv microbenchmark over the gn_nap0 option, then the quick c2 line per setting
for nap0 in 0 16 32 48; do for m in 1 2; do for SH in "1200 256 0 256 3 1 0" "75 1024 0 1024 3 1 0" "600 512 0 512 3 1 0"; do
  echo "== gnepi=$m nap0=$nap0 $SH"
  LDC_OPTIONS=gn_nap0=$nap0 LDC_MB_GNEPI=$m LDC_B=16 LDC_CONV_STAMPS=1 python tools/conv_one.py $SH 50 2>&1 | grep -v "XCC\|workgroup %\|stamps (s_mem"
done; done; done
for nap0 in 0 16 32 0 16 32; do
  LDC_OPTIONS=gn_nap0=$nap0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nap0=$nap0', round(d['value'],1), round(d['ms_per_step'],2))"
done
