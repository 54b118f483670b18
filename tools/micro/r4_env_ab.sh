#!/bin/bash
# round 4: same-box A/B of environment switches on the whole step (alternating, two rounds): "name=VAR=value" items, "base" = none.
#   bash tools/micro/r4_env_ab.sh <name> base s2p16=LP_S2P16=1 kernarg0=HIP_FORCE_DEV_KERNARG=0 [-- bench args]
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
NAME=$1; shift
ITEMS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ITEMS+=("$1"); shift; done
[ "$1" = "--" ] && shift
for rep in 1 2; do
for it in "${ITEMS[@]}"; do
  tag=${it%%=*}; kv=${it#*=}
  if [ "$it" = base ]; then pre=""; else pre="$kv"; fi
  env $pre timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline "$@" > gpurun_out/r4_${NAME}_bench_${tag}_$rep.json 2> gpurun_out/r4_${NAME}_bench_${tag}_$rep.err || { echo "$tag failed"; tail -3 gpurun_out/r4_${NAME}_bench_${tag}_$rep.err; continue; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4_${NAME}_bench_${tag}_$rep.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$tag', 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac_event', r.get('frac_event'), 'backbone', r.get('backbone_frac'), 'fwd_ms', r['forward_device_ms'])
PY
done
done
