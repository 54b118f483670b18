#!/bin/bash
# round-3 experiment 1 (no code change): single execution lane x batches in flight x HW queues
out=gpurun_out/r3_exp1; mkdir -p $out
line() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$1', 'value', d['value'], 'inflight1', d['value_inflight1'], 'step', d['step_ms'])
except Exception as e: print('$1', 'FAILED', e)"; }
run() { tag=$1; infl=$2; shift; shift; timeout -k 10 240 env "$@" python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --inflight $infl 2>$out/$tag.err | tee $out/$tag.json | line "$tag" | tee -a $out/summary.txt; }
for infl in 2 3 4 5 6 8 12; do run sl_i$infl $infl LP_SINGLE_LANE=1; done
for q in 1 3 5 6; do for infl in 3 4 6 8; do run sl_q${q}_i$infl $infl LP_SINGLE_LANE=1 GPU_MAX_HW_QUEUES=$q; done; done
for infl in 3 4 8 12; do run ml_i$infl $infl A=1; done
run sl_i6_nooverlap 6 LP_SINGLE_LANE=1 B=1
