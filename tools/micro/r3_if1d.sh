#!/bin/bash
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'step', d['step_ms'])"; }
for rep in 1 2 3; do
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 --single-lane 0 2>/dev/null | sec "yololps lanes"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 --single-lane 1 2>/dev/null | sec "yololps one_lane"
done
for rep in 1 2; do
python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 --single-lane 0 2>/dev/null | sec "yolov6m lanes"
python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 --single-lane 1 2>/dev/null | sec "yolov6m one_lane"
python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --inflight 1 --single-lane 0 2>/dev/null | sec "yololpn lanes"
python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --inflight 1 --single-lane 1 2>/dev/null | sec "yololpn one_lane"
python3 bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --inflight 1 --single-lane 0 2>/dev/null | sec "yololps_bs1 lanes"
python3 bench.py --batch 1 --steps 200 --warmup 20 --no-cpu-baseline --inflight 1 --single-lane 1 2>/dev/null | sec "yololps_bs1 one_lane"
done
