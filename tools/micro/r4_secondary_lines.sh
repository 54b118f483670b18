#!/bin/bash
# The bench lines of the two other single-GPU configurations, run AFTER their trace roofline / PMC traffic files are in profiles/ (the lines then carry
# them): overwrites gpurun_out/prof_<tag>/bench_*.json, per_op_*.txt and the first / third line of secondary_configs.txt.  bash tools/micro/r4_secondary_lines.sh r04
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/prof_$tag; mkdir -p $out
timeout -k 10 300 python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --detail $out/per_op_yololpn_bs128.txt > $out/bench_yololpn_bs128.json 2>/dev/null || exit 1
timeout -k 10 400 python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --detail $out/per_op_yolov6m_1280.txt > $out/bench_yolov6m_1280.json 2>/dev/null || exit 1
python3 - $out <<'P'
import json, sys
out = sys.argv[1]
def line(name, f):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    return '%s value %s value_inflight1 %s ms_per_step %s roofline %s %s %s frac %s traffic %s forward_device_ms %s nms_device_ms %s step_ms_inflight1 %s' % (
        name, d['value'], d['value_inflight1'], d['ms_per_step'], r['bound'], r['achieved'], r['unit'], r['frac'], r['traffic'], r['forward_device_ms'], r['nms_device_ms'], d['step_ms_inflight1'])
L = open(out + '/secondary_configs.txt').read().splitlines() if __import__('os').path.exists(out + '/secondary_configs.txt') else ['', '', '']
L[0] = line('yololpn 640 bs128 f16 (detections-only forward)', out + '/bench_yololpn_bs128.json')
L[2] = line('yolov6m 1280 bs8 bf16 (detections-only forward)', out + '/bench_yolov6m_1280.json')
open(out + '/secondary_configs.txt', 'w').write('\n'.join(L) + '\n')
print(L[0][:260]); print(L[2][:260])
P
