#!/bin/bash
# same-box A/B of two builds on the headline bench: libyololp_hip_prev.so against libyololp_hip.so
#   bash tools/micro/bench_ab.sh <tag> [bench args]
tag=${1:-x}; shift
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/bench_ab_$tag.txt
: > $out
for lib in yolo-lp_amd/libyololp_hip_prev.so yolo-lp_amd/libyololp_hip.so yolo-lp_amd/libyololp_hip_prev.so yolo-lp_amd/libyololp_hip.so; do
    n=$(basename $lib .so)
    LP_HIP_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --detail gpurun_out/bench_ab_${tag}_$n.txt "$@" > gpurun_out/bench_ab_$tag.json 2> gpurun_out/bench_ab_$tag.err || { echo "bench failed for $lib" >> $out; tail -5 gpurun_out/bench_ab_$tag.err >> $out; continue; }
    python3 -c "
import json;d=json.load(open('gpurun_out/bench_ab_$tag.json'));r=d['roofline']
print('$n', 'value', d['value'], 'inflight1', d['value_inflight1'], 'frac', r['frac'], 'avg3x3_ms', r.get('avg_launch_ms'), 'fwd_ms', r['forward_device_ms'], 'nms_ms', r['nms_device_ms'])" >> $out
done
cat $out
head -4 gpurun_out/bench_ab_${tag}_libyololp_hip_prev.txt
head -4 gpurun_out/bench_ab_${tag}_libyololp_hip.txt
