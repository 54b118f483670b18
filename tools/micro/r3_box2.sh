#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for cfg in "" "--model yolov6m --batch 8 --size 1280 --dtype bf16" "--model yololpn --batch 128"; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_box_kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 $cfg > /dev/null 2>&1
echo "== $cfg"; python3 $GRAFT_REPO_ROOT/tools/micro/step_kernels.py $GRAFT_REPO_ROOT/gpurun_out/r3_box_kt 20 | grep -E "head_|MODE|Li2ELi1ELi1ELi2|sum of"
rm -rf $GRAFT_REPO_ROOT/gpurun_out/r3_box_kt
done
