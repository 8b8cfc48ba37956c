for shape in "600 256 0 256 3 1 0" "300 512 256 512 3 1 0" "300 512 0 512 3 1 0" "600 256 0 384 1 1 0" "75 1024 0 1024 3 1 0"; do
  echo -n "$shape : "; LDC_B=${LDC_B:-32} python tools/conv_one.py $shape 50
done
