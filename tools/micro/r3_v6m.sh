#!/bin/bash
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'fwd_ms', d['roofline'].get('forward_device_ms'))"; }
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_gputest22.log 2>&1; tail -3 gpurun_out/r3_gputest22.log
for rep in 1 2; do
python3 _r02/bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | sec "r02 yolov6m"
python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --detail gpurun_out/r3_v6m_per_op.txt 2>/dev/null | sec "r03 yolov6m"
done
grep -E "pool|head_" gpurun_out/r3_v6m_per_op.txt
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r03 yololps"
