#!/bin/bash
# round 4: yolov6m 1280 bs8 bf16 with / without the 16x16x32 family, six batches and one batch in flight, alternating on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2 3; do
for m in 1 0; do
  if [ $m = 0 ]; then export LP_NO_MFMA16=1; else unset LP_NO_MFMA16; fi
  timeout -k 10 300 python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('mfma16=$m', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'ms_per_step', d['ms_per_step'], 'frac_event', r.get('frac_event', r['frac']), 'fwd_ms', r.get('forward_device_ms'))"
done
done
} > gpurun_out/r4_v6m_fam.txt 2>&1
cat gpurun_out/r4_v6m_fam.txt
