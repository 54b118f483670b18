#!/bin/bash
# Same-box A/B of the whole round: round 2's final tree (git archive 52c6cf9 -> _r02/, built there) against this tree, bench.py run
# alternately on one box.  Prints value / value_inflight1 / roofline.frac per run.
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'ms_per_step', d['ms_per_step'], 'frac', r.get('frac_event', r['frac']), 'fwd_ms', r.get('forward_device_ms'), 'nms_ms', r.get('nms_device_ms'))"; }
for rep in 1 2; do
  python3 _r02/bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r02 yololps_bs32"
  python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r03 yololps_bs32"
done
python3 _r02/bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "r02 yolov6m_1280_bs8"
python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "r03 yolov6m_1280_bs8"
python3 _r02/bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "r02 yololpn_bs128"
python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "r03 yololpn_bs128"
