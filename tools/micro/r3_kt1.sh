#!/bin/bash
# kernel trace, one batch in flight, single lane: every kernel's own duration
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3_kt1}; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --inflight 1 --single-lane 1 ${@:2} > $out/bench.json 2> $out/bench.err
cd $GRAFT_REPO_ROOT
python3 tools/micro/step_kernels.py $out/kt 30 | tee $out/step_kernels.txt
