for shape in "75 1024 0 1024 3 1 0" "150 512 0 512 3 1 0" "75 1024 1024 1024 3 1 0" "150 1024 0 1024 3 1 0"; do
 for B in 16 32; do
  echo "== $shape B=$B"
  echo -n "64x64   : "; LDC_B=$B LDC_CONV_SMALL_TILES=1000 python tools/conv_one.py $shape 50
  echo -n "128x64  : "; LDC_B=$B LDC_CONV_SMALL_TILES=0 LDC_CONV_MEDIUM_TILES=1000 python tools/conv_one.py $shape 50
  echo -n "128x128 : "; LDC_B=$B LDC_CONV_SMALL_TILES=0 LDC_CONV_MEDIUM_TILES=0 python tools/conv_one.py $shape 50
 done
done
