#!/bin/bash
# round 4: 16x16x32 pipelined kernel -- parity tests, then same-box A/B against the 32x32x16 variants per layer
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "pipe16 or two_destination or pipe_ring" > gpurun_out/r4_p16_test.log 2>&1; rc=$?
tail -15 gpurun_out/r4_p16_test.log
[ $rc -ne 0 ] && exit $rc
{
for rep in 1 2; do
for spec in "256 256 40 32 32" "256 256 40 32 39" "128 128 80 32 32" "128 128 80 32 39" "128 128 40 32 32" "128 128 40 32 39" "128 128 40 32 41" "256 256 20 32 34" "256 256 20 32 41" "256 256 20 32 39" "64 64 160 32 33" "64 64 160 32 40" "512 512 20 32 32" "512 512 20 32 39" "64 64 80 32 33" "64 64 80 32 40"; do
  set -- $spec
  sl=5; [ $3 -ge 80 ] && sl=3
  timeout -k 10 120 python tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch $4 --sl $sl --variant $5,3 2>&1 | tail -1
done
done
} > gpurun_out/r4_p16_convbench.log 2>&1
cat gpurun_out/r4_p16_convbench.log
