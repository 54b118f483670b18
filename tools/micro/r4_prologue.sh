#!/bin/bash
# round 4: what the prologue of the block-tiled kernel waits for -- stamp builds with / without its weight and halo pieces
# (make p16v TAG=st|stw|sth with -DLP_STAMPS [-DLP_P16V_PRO_SKIPW | -DLP_P16V_PRO_SKIPH]; the skip builds give wrong sums)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAGS=${@:-st stw sth}
{
for rep in 1 2; do
for tag in $TAGS; do
  export LP_HIP_LIB=$GRAFT_REPO_ROOT/yolo-lp_amd/libyololp_hip_p16v_$tag.so
  for spec in "256 256 40 32 42" "128 128 80 32 42"; do
    set -- $spec
    sl=5; [ $3 -ge 80 ] && sl=3
    echo "== $tag"; timeout -k 10 120 python tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch $4 --sl $sl --variant $5,3 --stamps 2>&1 | tail -4
  done
done
done
} > gpurun_out/r4_prologue.log 2>&1
cut -c1-260 gpurun_out/r4_prologue.log
