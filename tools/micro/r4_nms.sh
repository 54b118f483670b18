#!/bin/bash
# round 4: staged greedy NMS -- parity, then nms_device_ms of the secondary configurations with both forms
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hip_model.py tests/test_hip_kernels.py -m gpu -x -q -k "nms or bifusion" > gpurun_out/r4_nms_test.log 2>&1; rc=$?
tail -12 gpurun_out/r4_nms_test.log
[ $rc -ne 0 ] && exit $rc
for m in 1 0; do
  export LP_NMS_STAGED=$m
  timeout -k 10 400 python bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_nms_v6m_$m.json 2>/dev/null || exit 1
  timeout -k 10 300 python bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r4_nms_lpn_$m.json 2>/dev/null || exit 1
  python - <<PY
import json
for c in ('v6m','lpn'):
    d=json.loads(open('gpurun_out/r4_nms_%s_$m.json'%c).read().strip().splitlines()[-1]); r=d['roofline']
    print('staged=$m', c, 'value', d['value'], 'inflight1', d['value_inflight1'], 'nms_ms', r['nms_device_ms'], 'fwd_ms', r['forward_device_ms'])
PY
done
unset LP_NMS_STAGED
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --detail gpurun_out/r4_nms_perop.txt > gpurun_out/r4_nms_bench.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_nms_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('yololps', 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac_event', r['frac_event'], 'fwd_ms', r['forward_device_ms'], r['kernel'])
PY
sed -n 37,44p gpurun_out/r4_nms_perop.txt
