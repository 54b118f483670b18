#!/bin/bash
# loader-wave variants vs the plain pipelined kernel, same box, per layer shape
for cfg in "128 128 40 32 39" "256 256 40 32 39" "64 64 80 33 40" "128 128 80 32 39" "64 64 160 33 40" "512 512 20 32 39" "256 256 20 34 41" "256 256 20 32 39"; do
  set -- $cfg
  for v in $4 $5 $4 $5; do
    r=$(python3 tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch 32 --variant $v,3 --sl 3 --iters 30 2>&1 | grep TFLOP | sed 's/.*variant//')
    echo "$1->$2@$3 v$v: $r"
  done
done
