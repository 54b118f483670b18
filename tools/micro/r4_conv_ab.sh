#!/bin/bash
# round 4: block-tiled 16x16x32 kernel -- parity tests, then same-box A/B per layer against the fixed-tile variants
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
true > gpurun_out/r4_p16v_test.log 2>&1; rc=$?
tail -15 gpurun_out/r4_p16v_test.log
[ $rc -ne 0 ] && exit $rc
{
for rep in 1 2; do
for spec in "256 256 40 32 32" "256 256 40 32 39" "256 256 40 32 42" "128 128 80 32 32" "128 128 80 32 42" "128 128 40 32 32" "128 128 40 32 42" "128 128 40 32 43" "256 256 20 32 34" "256 256 20 32 42" "256 256 20 32 43" "64 64 160 32 33" "64 64 160 32 44" "512 512 20 32 32" "512 512 20 32 42" "512 512 20 32 43" "64 64 80 32 33" "64 64 80 32 44" "64 128 80 32 32" "64 128 80 32 42"; do
  set -- $spec
  sl=5; [ $3 -ge 80 ] && sl=3
  timeout -k 10 120 python tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch $4 --sl $sl --variant $5,3 2>&1 | tail -1
done
done
} > gpurun_out/r4_p16v_convbench.log 2>&1
cat gpurun_out/r4_p16v_convbench.log
