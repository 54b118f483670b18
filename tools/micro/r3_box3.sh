#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "det or detect or route or candidates or inflight or batch or config" > gpurun_out/r3_box3_tests.log 2>&1; tail -2 gpurun_out/r3_box3_tests.log
cd /tmp && export TMPDIR=/tmp
for cfg in "" "--model yololpn --batch 128"; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_hd_kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 $cfg > $GRAFT_REPO_ROOT/gpurun_out/r3_hd3.json 2>/dev/null
echo "== $cfg $(python3 -c "import json;d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/r3_hd3.json').read().strip().splitlines()[-1]);print(d['value'])")"; python3 $GRAFT_REPO_ROOT/tools/micro/step_kernels.py $GRAFT_REPO_ROOT/gpurun_out/r3_hd_kt 20 | grep -E "head_box|sum of" | cut -c1-110
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3_hd_kt
done
