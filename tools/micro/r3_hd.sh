#!/bin/bash
# head_det_kernel durations, product build vs stamps build, in isolation (single lane, default variants)
cd /tmp && export TMPDIR=/tmp
for lib in libyololp_hip.so libyololp_hip_stamps.so; do
  out=$GRAFT_REPO_ROOT/gpurun_out/r3_hd/$lib; mkdir -p $out
  LP_HIP_LIB=$GRAFT_REPO_ROOT/yolo-lp_amd/$lib timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -o kt -- python3 $GRAFT_REPO_ROOT/tools/micro/head_det_bench.py "$@" > $out/log.txt 2>&1
  python3 - $out <<'PY'
import csv,glob,sys,collections
d=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'head_det' in r['Kernel_Name'] or 'MODE' in r['Kernel_Name']: d[r['Kernel_Name'][:40]+' wgs%d'%(int(r['Grid_Size_X'])//192)].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in sorted(d.items()): print(sys.argv[1].split('/')[-1], k, 'n', len(v), 'median %.1f us' % sorted(v)[len(v)//2], 'min %.1f' % min(v))
PY
done
