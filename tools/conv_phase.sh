# Where a conv workgroup's time goes at the half-batch grid: phase stamps x (normal | no copies in the loop | no MFMAs) x tile
for shape in "1200 256 0 256 3 1 0" "300 512 0 512 3 1 0" "75 1024 0 1024 3 1 0" "1200 256 256 256 1 1 0"; do
 for tile in "64x64 LDC_CONV_SMALL_TILES=100000" "128x64 LDC_CONV_SMALL_TILES=0 LDC_CONV_MEDIUM_TILES=100000" "128x128 LDC_CONV_SMALL_TILES=0 LDC_CONV_MEDIUM_TILES=0"; do
  set -- $tile; name=$1; shift
  for dbg in 0 1 2; do
   echo "== $shape tile=$name debug=$dbg"
   env LDC_B=16 LDC_CONV_SPLITK=0 LDC_CONV_STAMPS=1 LDC_CONV_DEBUG=$dbg "$@" python tools/conv_one.py $shape 50 2>&1 | tail -2
  done
 done
done
