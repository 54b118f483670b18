#!/bin/bash
for lib in libyololp_hip.so libyololp_hip_oldpipe.so; do
  fails=0
  for i in 1 2 3 4 5 6 7 8; do
    LP_HIP_LIB=yolo-lp_amd/$lib timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -m gpu -q > gpurun_out/r3_stress3_$lib.$i.log 2>&1 || { fails=$((fails+1)); grep -E "^FAILED" gpurun_out/r3_stress3_$lib.$i.log | head -3; }
  done
  echo "$lib: $fails failing runs of 8"
done
