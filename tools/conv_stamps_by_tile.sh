# round 3: where a workgroup's time goes, tall tiles vs 128x64, B = 16 / 32 (stamps + ablations)
for B in 16 32; do
for shape in "1200 256 0 256 3 1 0" "300 512 0 512 3 1 0"; do
 for cfg in 0 1 2; do
  for dbg in 0 1 2; do
   echo "== B=$B $shape cfg=$cfg debug=$dbg"
   env LDC_B=$B LDC_TILE_CFG=$cfg LDC_CONV_SPLITK=0 LDC_CONV_STAMPS=1 LDC_CONV_DEBUG=$dbg python tools/conv_one.py $shape 50 2>&1 | tail -2
  done
 done
done
done
