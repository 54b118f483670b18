#!/bin/bash
# Timing-only ablations of the pipelined 3x3 kernel (make -C yolo-lp_amd/csrc abl): 1 no DMA, 2 no MFMA, 4 no fragment reads.
cd "$(dirname "$0")/../.."
for shape in "256 256 40 32" "64 64 160 33"; do
  set -- $shape
  for x in "" ${ABLS:-1 2 4 3 6}; do
    lib=yolo-lp_amd/libyololp_hip${x:+_abl$x}.so
    echo -n "abl=${x:-0} "; LP_HIP_LIB=$PWD/$lib python tools/conv_bench.py --batch 32 --k 3 --cin $1 --cout $2 --hw $3 --variant $4,3 --iters 30 2>&1 | tail -1
  done
done
