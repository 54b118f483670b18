#!/bin/bash
# round 4: GPU suite + per-layer reference timings + bench line on one box
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-base}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_${TAG}_gputest.log 2>&1; rc=$?
tail -5 gpurun_out/r4_${TAG}_gputest.log
[ $rc -ne 0 ] && exit $rc
{
for spec in "256 256 40 32 32" "128 128 80 32 32" "128 128 40 32 32" "128 128 40 32 34" "256 256 20 32 32" "256 256 20 32 34" "64 64 160 32 33" "512 512 20 32 32" "64 64 80 32 33"; do
  set -- $spec
  sl=5; [ $3 -ge 80 ] && sl=3
  timeout -k 10 120 python tools/conv_bench.py --cin $1 --cout $2 --hw $3 --batch $4 --sl $sl --variant $5,3 2>&1 | tail -1
done
} > gpurun_out/r4_${TAG}_convbench.log 2>&1
cat gpurun_out/r4_${TAG}_convbench.log
timeout -k 10 600 python bench.py > gpurun_out/r4_${TAG}_bench.json 2> gpurun_out/r4_${TAG}_bench.err; rc=$?
tail -c 1500 gpurun_out/r4_${TAG}_bench.json
exit $rc
