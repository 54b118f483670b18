#!/bin/bash
timeout -k 10 600 python -m pytest tests -m gpu -q -x -k "stem or fused" > gpurun_out/r3_stem_tests.log 2>&1; tail -2 gpurun_out/r3_stem_tests.log
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', d['value'], 'inflight1', d.get('value_inflight1'))"; }
for rep in 1 2; do
LP_HIP_LIB=yolo-lp_amd/libyololp_hip_oldstem.so python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --detail gpurun_out/r3_stem_old.txt 2>/dev/null | sec "old"
python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --detail gpurun_out/r3_stem_new.txt 2>/dev/null | sec "new"
done
sed -n 3p gpurun_out/r3_stem_old.txt; sed -n 3p gpurun_out/r3_stem_new.txt
LP_HIP_LIB=yolo-lp_amd/libyololp_hip_oldstem.so python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --detail gpurun_out/r3_stem_old.txt 2>/dev/null | sec "old lpn"
python3 bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --detail gpurun_out/r3_stem_new.txt 2>/dev/null | sec "new lpn"
sed -n 3p gpurun_out/r3_stem_old.txt; sed -n 3p gpurun_out/r3_stem_new.txt
