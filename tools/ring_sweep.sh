export LDC_B=16
for shape in "75 1024 0 1024 3 1 0" "150 512 0 512 3 1 0" "75 1024 1024 1024 3 1 0" "75 1024 1024 1024 1 1 0"; do
 for S in 2 3 4; do echo -n "small $shape S=$S: "; LDC_CONV_RING_S=$S python tools/conv_one.py $shape 50; done
done
for shape in "1200 256 0 256 3 1 0" "300 512 0 512 3 1 0" "600 512 256 512 3 1 0" "1200 256 256 256 1 1 0" "150 1024 0 1024 3 1 0"; do
 for S in 2 3; do echo -n "medium $shape S=$S: "; LDC_CONV_RING_M=$S python tools/conv_one.py $shape 50; done
done
