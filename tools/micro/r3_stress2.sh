#!/bin/bash
for lib in libyololp_hip.so libyololp_hip_oldpipe.so; do
  echo "== $lib"; LP_HIP_LIB=yolo-lp_amd/$lib timeout -k 10 300 python3 tools/micro/pipe_stress.py 2>&1 | tail -8
  LP_HIP_LIB=yolo-lp_amd/$lib timeout -k 10 300 python3 tools/micro/pipe_stress.py 16 24 33 47 2 35 300 2>&1 | tail -4
  LP_HIP_LIB=yolo-lp_amd/$lib timeout -k 10 300 python3 tools/micro/pipe_stress.py 128 128 40 40 8 32 200 2>&1 | tail -4
done
