#!/bin/bash
# round-3 experiment 2 (no code change): single lane, six batches in flight: kernel footprint / grid caps
out=gpurun_out/r3_exp2; mkdir -p $out
line() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$1', 'value', d['value'], 'inflight1', d['value_inflight1'], 'step', d['step_ms'])
except Exception as e: print('$1', 'FAILED', e)"; }
run() { tag=$1; infl=$2; shift; shift; timeout -k 10 240 env "$@" python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --inflight $infl 2>$out/$tag.err | tee $out/$tag.json | line "$tag" | tee -a $out/summary.txt; }
run sl_i6 6 LP_SINGLE_LANE=1
run sl_nopipe_i6 6 LP_SINGLE_LANE=1 LP_NO_PIPE=1
run sl_maxwg128_i6 6 LP_SINGLE_LANE=1 LP_PIPE_MAXWG=128
run sl_maxwg192_i6 6 LP_SINGLE_LANE=1 LP_PIPE_MAXWG=192
run sl_maxwg128_i8 8 LP_SINGLE_LANE=1 LP_PIPE_MAXWG=128
run sl_viapred_i6 6 LP_SINGLE_LANE=1
run sl_nostream_i6 6 LP_SINGLE_LANE=1 LP_NO_STREAM=1
run sl_nofused_i6 6 LP_SINGLE_LANE=1 LP_NO_FUSED_STEM=1 LP_NO_FUSED_PW=1
run sl_i6_b 6 LP_SINGLE_LANE=1
