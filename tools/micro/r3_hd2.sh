#!/bin/bash
cd /tmp && export TMPDIR=/tmp
for pc in 3 2 1; do
  out=$GRAFT_REPO_ROOT/gpurun_out/r3_hd2/pc$pc; mkdir -p $out
  LP_DET_PERCU=$pc timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 $GRAFT_REPO_ROOT/tools/micro/head_det_bench.py "$@" > $out/log.txt 2>&1
  python3 - $out $pc <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'head_det' in r['Kernel_Name']: d[r['Kernel_Name'][18:40]+' wgs%d'%(int(r['Grid_Size_X'])//192)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()): print('per_cu', sys.argv[2], k, 'median %.1f us' % sorted(v)[len(v)//2])
PY
done
