#!/bin/bash
# conv3x3_s2p16_kernel with every halo piece fetched as one contiguous KiB (timing only, wrong sums: P16V="-DLP_S2_ABL_DENSE" make p16v TAG=s2dense)
# against the real gather (32 segments of 32 bytes per piece): what the NHWC halo costs a stride-2 layer
cd "$(dirname "$0")/../.."
run() { python tools/conv_bench.py --batch 32 --k 3 --s 2 --cin $1 --cout $2 --hw $3 --sl $4 --variant $5 --iters 30 2>&1 | tail -1; }
for lib in "" yolo-lp_amd/libyololp_hip_p16v_s2dense.so; do
  echo "== LP_HIP_LIB=$lib"
  for l in "64 128 160 2" "128 256 80 3" "256 512 40 4" "128 128 80 3"; do
    set -- $l
    if [ -n "$lib" ]; then LP_HIP_LIB=$PWD/$lib run $1 $2 $3 $4 48,3; else run $1 $2 $3 $4 48,3; fi
  done
done
