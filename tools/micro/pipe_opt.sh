#!/bin/bash
# Experiment builds of the pipelined 3x3 kernel (make -C yolo-lp_amd/csrc opt / abl): every libyololp_hip_{opt,abl}*.so present.
# SHAPES="cin_cout_hw_variant ..." (variant 32 = PIPE_D, 33 = PIPE_B, 34 = PIPE_F)
cd "$(dirname "$0")/../.."
for shape in ${SHAPES:-256_256_40_32}; do
  set -- ${shape//_/ }
  for lib in yolo-lp_amd/libyololp_hip.so yolo-lp_amd/libyololp_hip_abl*.so yolo-lp_amd/libyololp_hip_opt*.so; do
    [ -f $lib ] || continue
    echo -n "$(basename $lib) "; LP_HIP_LIB=$PWD/$lib python tools/conv_bench.py --batch 32 --k 3 --cin $1 --cout $2 --hw $3 --variant $4,3 --iters 30 2>&1 | tail -1
  done
done
