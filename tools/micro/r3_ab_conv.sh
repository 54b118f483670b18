#!/bin/bash
# same-box A/B of builds ($LIBS) on the 3x3 layer shapes of yololps (pipelined variants)
for cfg in "128 128 40 32" "256 256 40 32" "64 64 80 33" "128 128 80 32" "64 64 160 33" "512 512 20 32" "256 256 20 34"; do
  set -- $cfg
  for lib in ${LIBS}; do
    r=$(LP_HIP_LIB=yolo-lp_amd/$lib python3 tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch 32 --variant $4,3 --sl 3 --iters 30 2>&1 | grep TFLOP | sed 's/.*variant//')
    echo "$1->$2@$3 $lib: $r"
  done
done
