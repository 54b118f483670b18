#!/bin/bash
# same-box A/B of builds ($LIBS) on the 1x1 (streaming kernel) layer shapes of yololps
for cfg in "256 128 40 1 1 16,2" "256 128 40 1 1 17,2" "512 256 20 1 1 17,2" "128 64 80 1 1 16,2" "384 128 40 1 1 17,2" "192 64 80 1 1 16,2" "64 64 160 1 1 16,2"; do
  set -- $cfg
  for lib in ${LIBS}; do
    r=$(LP_HIP_LIB=yolo-lp_amd/$lib python3 tools/conv_bench.py --cin $1 --cout $2 --hw $3 --k $4 --s $5 --batch 32 --variant $6 --sl 2 --iters 30 2>&1 | grep -E "TFLOP|rror" | sed 's/.*variant//')
    echo "$1->$2@$3 k$4 s$5 $lib: $r"
  done
done
