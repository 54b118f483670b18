#!/bin/bash
# The stride-2 3x3 layers (B=32 unless noted) on conv3x3_s2p16_kernel (variants 48 / 49) against conv_mfma_kernel<KS=3,S=2> (the default; its tiles)
cd "$(dirname "$0")/../.."
run() { python tools/conv_bench.py --batch ${6:-32} --k 3 --s 2 --cin $1 --cout $2 --hw $3 --sl $4 --variant $5 --iters 30 --dtype ${7:-f16} 2>&1 | tail -1; }
for l in "64 128 160 2" "128 256 80 3" "256 512 40 4" "128 128 80 3" "128 128 40 4"; do
  set -- $l
  for v in 3,2 3,1 5,1 5,2; do run $1 $2 $3 $4 $v; done
  run $1 $2 $3 $4 48,3; run $1 $2 $3 $4 49,3
done
# yolov6m 1280x1280 bs=8 bf16
for l in "48 96 640 1" "96 192 320 2" "192 384 160 3" "384 768 80 4" "96 96 320 2" "192 192 160 3"; do
  set -- $l
  for v in 3,2 5,1 4,1; do run $1 $2 $3 $4 $v 8 bf16; done
  run $1 $2 $3 $4 48,3 8 bf16; run $1 $2 $3 $4 49,3 8 bf16
done
