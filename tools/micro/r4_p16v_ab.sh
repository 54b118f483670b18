#!/bin/bash
# round 4: same-box A/B of experiment builds of the block-tiled kernel (make p16v TAG=...), alternating libraries
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAGS=${@:-"b15 b36 b16 prio"}
{
for rep in 1 2 3; do
for tag in base $TAGS; do
  if [ $tag = base ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/yolo-lp_amd/libyololp_hip_p16v_$tag.so; fi
  for spec in "256 256 40 32 42" "128 128 80 32 42" "128 256 40 32 42"; do
    set -- $spec
    sl=5; [ $3 -ge 80 ] && sl=3
    echo -n "$tag  "; timeout -k 10 120 python tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch $4 --sl $sl --variant $5,3 2>&1 | tail -1
  done
done
done
} > gpurun_out/r4_p16v_ab.log 2>&1
cat gpurun_out/r4_p16v_ab.log | cut -c1-110
