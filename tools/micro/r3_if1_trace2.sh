#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3_if1_trace2; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export LP_SINGLE_LANE=1
run() { # name, tree, extra args
  rocprofv3 --kernel-trace --output-format csv -d $out/kt_$1 -o kt -- python3 $root/$2bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 --via-pred $3 > $out/$1.json 2> $out/$1.err
  python3 $root/tools/micro/step_kernels.py $out/kt_$1 20 > $out/step_$1.txt; rm -rf $out/kt_$1; tail -1 $out/step_$1.txt
}
run r02 _r02/ ""
LP_NO_SIBLINGS=1 run r03_nosib "" "--single-lane 1"
run r03 "" "--single-lane 1"
run r02b _r02/ ""
