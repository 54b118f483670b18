#!/bin/bash
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'), 'step', d['step_ms'])"; }
for rep in 1 2 3; do
python3 _r02/bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r02"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | sec "r03"
done
python3 _r02/bench.py --steps 50 --warmup 5 --no-cpu-baseline --via-pred 2>/dev/null | sec "r02_viapred"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --via-pred 2>/dev/null | sec "r03_viapred"
