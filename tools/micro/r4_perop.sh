#!/bin/bash
# round 4: bench line + per-op table (optionally with LP_NO_MFMA16=1 for the same-box A/B of the MFMA family rule)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
TAG=${1:-perop}
for m in 1 0; do
  if [ $m = 0 ]; then export LP_NO_MFMA16=1; else unset LP_NO_MFMA16; fi
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --detail gpurun_out/r4_${TAG}_perop_m$m.txt > gpurun_out/r4_${TAG}_bench_m$m.json 2> gpurun_out/r4_${TAG}_bench_m$m.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_${TAG}_bench_m$m.json').read().strip().splitlines()[-1]); r=d['roofline']
print('mfma16=$m', 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac_event', r['frac_event'], 'fwd_ms', r['forward_device_ms'])
PY
done
