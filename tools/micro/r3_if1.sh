#!/bin/bash
# one batch in flight: round 2's tree (_r02/) against this tree with this round's graph-level changes switched off one by one
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'step', d['step_ms'])"; }
for rep in 1 2; do
python3 _r02/bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r02"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03"
LP_NO_SIBLINGS=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_nosiblings"
LP_NO_HEAD_DET=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_noheaddet"
LP_NO_SIBLINGS=1 LP_NO_HEAD_DET=1 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 2>/dev/null | sec "r03_neither"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 --single-lane 1 2>/dev/null | sec "r03_singlelane"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 --via-pred 2>/dev/null | sec "r03_viapred"
python3 _r02/bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 --via-pred 2>/dev/null | sec "r02_viapred"
done
