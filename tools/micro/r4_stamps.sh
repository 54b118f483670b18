#!/bin/bash
# round 4: in-kernel stamps of the block-tiled kernel (diagnostic build: libyololp_hip_stamps.so, never the product path)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp LP_HIP_LIB=$GRAFT_REPO_ROOT/yolo-lp_amd/libyololp_hip_stamps.so
mkdir -p gpurun_out
{
for spec in "256 256 40 32 42" "256 256 40 32 32" "128 128 80 32 42" "128 128 40 32 39" "128 128 40 32 43" "256 256 20 32 41"; do
  set -- $spec
  sl=5; [ $3 -ge 80 ] && sl=3
  timeout -k 10 120 python tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch $4 --sl $sl --variant $5,3 --stamps 2>&1 | tail -5
done
} > gpurun_out/r4_stamps.log 2>&1
cat gpurun_out/r4_stamps.log
