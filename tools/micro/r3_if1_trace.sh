#!/bin/bash
# same-box kernel traces, one batch in flight, one lane, through the prediction tensor: round 2's tree against this one
out=$GRAFT_REPO_ROOT/gpurun_out/r3_if1_trace; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export LP_SINGLE_LANE=1
rocprofv3 --kernel-trace --output-format csv -d $out/kt_r02 -o kt -- python3 $root/_r02/bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 --via-pred > $out/r02.json 2> $out/r02.err
rocprofv3 --kernel-trace --output-format csv -d $out/kt_r03 -o kt -- python3 $root/bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 --single-lane 1 --via-pred > $out/r03.json 2> $out/r03.err
python3 $root/tools/micro/step_kernels.py $out/kt_r02 20 > $out/step_r02.txt
python3 $root/tools/micro/step_kernels.py $out/kt_r03 20 > $out/step_r03.txt
rm -rf $out/kt_r02 $out/kt_r03
tail -1 $out/step_r02.txt; tail -1 $out/step_r03.txt
