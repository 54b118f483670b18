#!/bin/bash
out=gpurun_out/r3_run5; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_hip_model.py -m gpu -q -k "detections_only or batch or bs32 or bs128 or bs8 or inflight or switches" > $out/gputest.log 2>&1; tail -4 $out/gputest.log
bash tools/micro/r3_kt1.sh r3_run5/kt1 > $out/kt1.log 2>&1; grep -E "head_det|conv_mfma_kernelIDF16_Li2ELi1|sum of" $out/kt1.log
sec() { python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
    print('$1', 'value', d['value'], 'inflight1', d['value_inflight1'], 'ms_per_step', d['ms_per_step'], 'roofline', r['bound'], r['achieved'], r['frac'], 'fwd_ms', r['forward_device_ms'], 'nms_ms', r['nms_device_ms'])
except Exception as e: print('$1 FAILED', e)"; }
timeout -k 10 300 python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --detail $out/per_op_yololpn.txt 2>$out/lpn.err | tee $out/lpn.json | sec "yololpn bs128"
timeout -k 10 400 python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --detail $out/per_op_v6m.txt 2>$out/v6m.err | tee $out/v6m.json | sec "yolov6m 1280 bs8 bf16"
