#!/bin/bash
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'step', d['step_ms'])"; }
for rep in 1 2; do
python3 _r02/bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r02"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03"
LP_EXP_NO_LEAVE=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_noleave"
LP_EXP_NO_DONE=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_nodone"
LP_EXP_NO_LEAVE=1 LP_EXP_NO_DONE=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_neither"
LP_EXP_NO_LEAVE=1 LP_EXP_NOFENCE=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_noleave_nofence"
LP_EXP_NO_LEAVE=1 LP_EXP_NO_DONE=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r03_neither_6inflight"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r03_6inflight"
done
