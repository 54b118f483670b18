#!/bin/bash
# round-3 experiment 0 (no code change): how much do batches in flight overlap, by kernel footprint and by HW-queue count?
out=gpurun_out/r3_exp0; mkdir -p $out
line() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('$1', 'value', d['value'], 'inflight1', d['value_inflight1'], 'fwd_dev_ms', r['forward_device_ms'], '3x3', r['achieved'], 'step', d['step_ms'], 'step1', d['step_ms_inflight1'])
except Exception as e: print('$1', 'FAILED', e)"; }
run() { tag=$1; shift; echo "== $tag"; timeout -k 10 240 env "$@" python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>$out/$tag.err | tee $out/$tag.json | line "$tag" | tee -a $out/summary.txt; }
run base A=1
run nopipe LP_NO_PIPE=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run single_lane LP_SINGLE_LANE=1
run single_lane_hwq8 LP_SINGLE_LANE=1 GPU_MAX_HW_QUEUES=8
run nopipe_hwq8 LP_NO_PIPE=1 GPU_MAX_HW_QUEUES=8
