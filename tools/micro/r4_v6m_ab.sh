#!/bin/bash
# round 4: yolov6m 1280 bs8 bf16, round 3's tree (_r03) against this tree with and without the 16x16x32 family, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
one() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'ms_per_step', d['ms_per_step'], 'fwd_ms', r.get('forward_device_ms'), 'nms_ms', r.get('nms_device_ms'))"; }
A="--model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 30 --warmup 3 --no-cpu-baseline"
{
for rep in 1 2 3; do
  unset LP_NO_MFMA16
  timeout -k 10 300 python3 _r03/bench.py $A 2>/dev/null | one r03
  timeout -k 10 300 python3 bench.py $A 2>/dev/null | one r04
  export LP_NO_MFMA16=1
  timeout -k 10 300 python3 bench.py $A 2>/dev/null | one r04_nomfma16
done
} > gpurun_out/r4_v6m_ab.txt 2>&1
cat gpurun_out/r4_v6m_ab.txt
