#!/bin/bash
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/r3_gputest25.log 2>&1; tail -2 gpurun_out/r3_gputest25.log
for cfg in "" "--model yololpn --batch 128" "--model yolov6m --batch 8 --size 1280 --dtype bf16"; do
python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --via-pred $cfg --detail gpurun_out/r3_rows_per_op.txt 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg via-pred value', d['value'], 'inflight1', d.get('value_inflight1'))"
grep head_cls gpurun_out/r3_rows_per_op.txt | cut -c1-100
done
