#!/bin/bash
# Same-box A/B of a whole round: the final tree of the round before (git archive <commit> | tar -x -C _r03; make there) against this
# tree, bench.py run alternately on one box.  Usage: round_ab.sh <reference dir> <its label> <this tree's label>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
REF=${1:-_r03}; LA=${2:-r03}; LB=${3:-r04}
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'ms_per_step', d['ms_per_step'], 'frac', r.get('frac_event', r['frac']), 'fwd_ms', r.get('forward_device_ms'), 'nms_ms', r.get('nms_device_ms'))"; }
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  timeout -k 10 300 python3 $REF/bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "$LA yololps_bs32"
  timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "$LB yololps_bs32"
done
for rep in 1 2; do
  timeout -k 10 300 python3 $REF/bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "$LA yolov6m_1280_bs8"
  timeout -k 10 300 python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "$LB yolov6m_1280_bs8"
  timeout -k 10 300 python3 $REF/bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "$LA yololpn_bs128"
  timeout -k 10 300 python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "$LB yololpn_bs128"
done
} > gpurun_out/round_ab_$LB.txt 2>&1
cat gpurun_out/round_ab_$LB.txt
