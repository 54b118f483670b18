#!/bin/bash
out=gpurun_out/r3_run3; mkdir -p $out
timeout -k 10 600 python -m pytest tests -m gpu -q > $out/gputest.log 2>&1; tail -8 $out/gputest.log
grep -i "rounding" gpurun_out/parity.log | tail -12
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --detail $out/per_op.txt > $out/bench.json 2> $out/bench.err; tail -c 2500 $out/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/kt6 -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/bench_kt6.json 2> $GRAFT_REPO_ROOT/$out/bench_kt6.err
cd $GRAFT_REPO_ROOT; ls -la $out/kt6/* | head; python3 tools/micro/kstats.py $out/kt6 | head -30
