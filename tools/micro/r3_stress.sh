#!/bin/bash
# repeat the pipelined-kernel bit-identity tests to expose an intermittent race
for lib in libyololp_hip.so libyololp_hip_oldpipe.so; do
  fails=0
  for i in 1 2 3 4 5 6; do
    LP_HIP_LIB=yolo-lp_amd/$lib timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -m gpu -q -k "conv3x3_pipe or every_kernel_variant" > gpurun_out/r3_stress_$lib.$i.log 2>&1 || fails=$((fails+1))
    tail -1 gpurun_out/r3_stress_$lib.$i.log
  done
  echo "$lib: $fails failing runs of 6"
done
