#!/bin/bash
# round 4: box predictors for the candidates only (head_box_det_kernel, sparse form) -- model tests, then a same-box A/B of the whole step
# against LP_NO_BOX_SPARSE=1 (boxes of every anchor), alternating; yololps default line and the dense-candidate yololpn recipe
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py -m gpu -x -q -k "det or head or nms or golden" > gpurun_out/r4_box_test.log 2>&1; rc=$?
tail -3 gpurun_out/r4_box_test.log
[ $rc -ne 0 ] && exit $rc
one() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'ms_per_step', d['ms_per_step'], 'device_ms', d['config']['device_ms_per_step'], 'cand', d['config']['mean_detections_per_image'])"; }
{
for rep in 1 2 3; do
  unset LP_NO_BOX_SPARSE
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | one "sparse lps"
  export LP_NO_BOX_SPARSE=1
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | one "dense  lps"
done
for rep in 1 2; do
  unset LP_NO_BOX_SPARSE
  timeout -k 10 300 python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | one "sparse lpn"
  export LP_NO_BOX_SPARSE=1
  timeout -k 10 300 python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | one "dense  lpn"
done
} > gpurun_out/r4_box_ab.txt 2>&1
cat gpurun_out/r4_box_ab.txt
