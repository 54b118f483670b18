#!/bin/bash
# round 4: fused BiFusion kernel -- parity test, then bench with per-op table (LP_NO_FUSED_BF=1 for the A/B)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "bifusion" > gpurun_out/r4_bf_test.log 2>&1; rc=$?
tail -25 gpurun_out/r4_bf_test.log
[ $rc -ne 0 ] && exit $rc
for m in 1 0; do
  if [ $m = 0 ]; then export LP_NO_FUSED_BF=1; else unset LP_NO_FUSED_BF; fi
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --detail gpurun_out/r4_bf_perop_m$m.txt > gpurun_out/r4_bf_bench_m$m.json 2> gpurun_out/r4_bf_bench_m$m.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_bf_bench_m$m.json').read().strip().splitlines()[-1]); r=d['roofline']
print('fused_bf=$m', 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac_event', r['frac_event'], 'fwd_ms', r['forward_device_ms'])
PY
  sed -n 37,44p gpurun_out/r4_bf_perop_m$m.txt
done
