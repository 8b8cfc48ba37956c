export LDC_B=16
for shape in "75 1024 0 1024 3 1 0" "75 1024 1024 1024 3 1 0" "75 1024 1024 1024 1 1 0" "150 512 0 512 3 1 0" "150 1024 0 1024 3 1 0" "75 1024 0 384 1 1 0" "300 512 0 512 3 1 0"; do
 for k in 0 2 3; do echo -n "$shape SPLITK=$k: "; LDC_CONV_SPLITK=$k python tools/conv_one.py $shape 50; done
done
