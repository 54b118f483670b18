#!/bin/bash
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'))"; }
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "det or detect or route or candidates or inflight" > gpurun_out/r3_box_tests.log 2>&1; tail -3 gpurun_out/r3_box_tests.log
for rep in 1 2; do
LP_NO_BOX_STREAM=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "generic_decode"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "box_stream"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_box_kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/micro/step_kernels.py $GRAFT_REPO_ROOT/gpurun_out/r3_box_kt 20 | grep -E "head_|sum of"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3_box_kt
