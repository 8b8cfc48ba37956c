# round 3: launch-level ablations (debug bits: 1 no copies in loop, 4 no output stores, 8 return at once, 16 one unit only)
for B in 16 32; do
for shape in "1200 256 0 256 3 1 0" "75 1024 0 1024 3 1 0"; do
 for cfg in 1 2; do
  for dbg in 0 4 8 16 20; do
   echo "== B=$B $shape cfg=$cfg debug=$dbg  $(env LDC_B=$B LDC_TILE_CFG=$cfg LDC_CONV_SPLITK=0 LDC_CONV_DEBUG=$dbg python tools/conv_one.py $shape 50 2>&1 | tail -1)"
  done
 done
done
done
