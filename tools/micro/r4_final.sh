#!/bin/bash
# round 4: the round-end checks on one box -- full GPU suite, smoke(), the default bench line (trace roofline + PMC traffic attached by hash),
# the two secondary lines (their own profile files)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4_final_gputest.log 2>&1; rc=$?
tail -3 gpurun_out/r4_final_gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_final_smoke.log 2>&1 || { tail -5 gpurun_out/r4_final_smoke.log; exit 1; }
tail -1 gpurun_out/r4_final_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r4_final_bench.json 2> gpurun_out/r4_final_bench.err || exit 1
timeout -k 10 300 python bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4_final_bench_v6m.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/r4_final_bench_lpn.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ('', '_v6m', '_lpn'):
    d = json.loads(open('gpurun_out/r4_final_bench%s.json' % f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f or 'lps', 'value', d['value'], 'inflight1', d['value_inflight1'], 'bound', r['bound'], 'frac', r['frac'], 'frac_event', r.get('frac_event'),
          'backbone_frac', r.get('backbone_frac'), 'traffic', r['traffic'], '|', r['frac_source'][:90], '|', r['traffic_source'][:60])
PY
