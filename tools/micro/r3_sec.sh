#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r3_sec; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o kt -- python3 $GRAFT_REPO_ROOT/bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --inflight 1 --single-lane 1 > $out/v6m.json 2> $out/v6m.err
cd $GRAFT_REPO_ROOT
python3 - $out <<'PY'
import csv,glob,sys,collections,json
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/kt/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        n=r['Kernel_Name']
        for key in ('greedy_kernel','sort_kernel','score_kernel','head_det_kernel'):
            if key in n: d[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print(k, 'n', len(v), 'median %.1f us' % sorted(v)[len(v)//2])
j=json.loads(open(sys.argv[1]+'/v6m.json').read().strip().splitlines()[-1]); print('yolov6m traced run: value', j['value'], 'inflight1', j['value_inflight1'])
PY
timeout -k 10 400 python3 bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --detail $out/per_op_v6m.txt | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('yolov6m', d['value'], d['value_inflight1'], r['forward_device_ms'], r['nms_device_ms'], r['frac'])"
