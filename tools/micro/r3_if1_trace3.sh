#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3_if1_trace3; mkdir -p $out; root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, tree, extra args
  rocprofv3 --kernel-trace --output-format csv -d $out/kt_$1 -o kt -- python3 $root/$2bench.py --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 $3 > $out/$1.json 2> $out/$1.err
  echo "== $1 $(python3 -c "import json;d=json.loads(open('$out/$1.json').read().strip().splitlines()[-1]);print(d['value'])")"; python3 $root/tools/micro/fwd_idle.py $out/kt_$1 20; rm -rf $out/kt_$1
}
run r02 _r02/ ""
run r03 "" ""
run r02_viapred _r02/ "--via-pred"
run r03_viapred "" "--via-pred"
run r03_sl "" "--single-lane 1"
