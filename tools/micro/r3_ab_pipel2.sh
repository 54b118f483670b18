#!/bin/bash
for cfg in "128 128 40 32 39" "256 256 40 32 39" "64 64 80 33 40" "128 128 80 32 39"; do
  set -- $cfg
  for lib in libyololp_hip.so libyololp_hip_nl4.so libyololp_hip_nl8.so; do
    for v in $4 $5; do
      [ $lib != libyololp_hip.so ] && [ $v = $4 ] && continue
      r=$(LP_HIP_LIB=yolo-lp_amd/$lib python3 tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch 32 --variant $v,3 --sl 3 --iters 30 2>&1 | grep TFLOP | sed 's/.*variant//')
      echo "$1->$2@$3 $lib v$v: $r"
    done
  done
done
