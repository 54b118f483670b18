#!/bin/bash
# round 4: kernel tests of the 3x3 families, then the secondary configurations' bench lines + per-op tables
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-sec}
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "pipe or two_destination or variant" > gpurun_out/r4_${TAG}_test.log 2>&1; rc=$?
tail -4 gpurun_out/r4_${TAG}_test.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --detail gpurun_out/r4_${TAG}_perop_v6m.txt > gpurun_out/r4_${TAG}_bench_v6m.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --detail gpurun_out/r4_${TAG}_perop_lpn.txt > gpurun_out/r4_${TAG}_bench_lpn.json 2>/dev/null || exit 1
python - <<PY
import json
for m in ('v6m','lpn'):
    d=json.loads(open('gpurun_out/r4_${TAG}_bench_%s.json'%m).read().strip().splitlines()[-1]); r=d['roofline']
    print(m, 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac', r['frac'], 'fwd_ms', r['forward_device_ms'], 'nms_ms', r['nms_device_ms'])
PY
