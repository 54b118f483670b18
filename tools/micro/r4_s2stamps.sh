#!/bin/bash
# in-kernel clock stamps of conv3x3_s2p16_kernel (make -C yolo-lp_amd/csrc stamps): where a chunk iteration's clocks go
cd "$(dirname "$0")/../.."
export LP_HIP_LIB=$PWD/yolo-lp_amd/libyololp_hip_stamps.so
for l in "64 128 160 2" "128 256 80 3" "256 512 40 4"; do
  set -- $l
  python tools/conv_bench.py --batch 32 --k 3 --s 2 --cin $1 --cout $2 --hw $3 --sl $4 --variant 48,3 --iters 10 --stamps 2>&1 | grep -v amdgpu.ids
done
