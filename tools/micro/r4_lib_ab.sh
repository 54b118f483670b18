#!/bin/bash
# round 4: same-box A/B of experiment builds of the library (make p16v TAG=...) on the whole step: bench line + per-op table,
# alternating libraries.  Usage: r4_lib_ab.sh <name> <tag> [<tag> ...]   ("base" = the product library)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NAME=$1; shift
TAGS=${@:-base}
for rep in 1 2; do
for tag in $TAGS; do
  if [ $tag = base ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/yolo-lp_amd/libyololp_hip_p16v_$tag.so; fi
  timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --detail gpurun_out/r4_${NAME}_perop_${tag}_$rep.txt > gpurun_out/r4_${NAME}_bench_${tag}_$rep.json 2> gpurun_out/r4_${NAME}_bench_${tag}_$rep.err || exit 1
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_${NAME}_bench_${tag}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac_event', r['frac_event'], 'backbone', r.get('backbone_frac'), 'fwd_ms', r['forward_device_ms'])
PY
done
done
