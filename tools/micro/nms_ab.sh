#!/bin/bash
# same-box A/B of the post-processing kernels: the committed build against libyololp_hip_prev.so
set -e
tag=${1:-x}
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/nms_ab_$tag.txt
: > $out
run() {
    name=$1; lib=$2; shift; shift
    rm -rf gpurun_out/kt_nms
    LP_HIP_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_nms -o kt -- python3 bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 > gpurun_out/kt_nms.json 2> gpurun_out/kt_nms.err
    echo "== $name ($lib)" >> $out
    python3 -c "import json;d=json.load(open('gpurun_out/kt_nms.json'));print('value_inflight1',d['value_inflight1'],'nms_device_ms',d['roofline'].get('nms_device_ms'))" >> $out
    python3 tools/micro/kstats.py gpurun_out/kt_nms | grep -i "score_kernel\|sort_kernel\|greedy_kernel\|head_cls_rows" >> $out || true
}
for lib in yolo-lp_amd/libyololp_hip_prev.so yolo-lp_amd/libyololp_hip.so; do
run "yololps bs32 f16 det" $lib
run "yololpn bs128 f16 via-pred" $lib --model yololpn --batch 128 --via-pred
run "yololpn bs128 f16 det" $lib --model yololpn --batch 128
run "yolov6m 1280 bs8 bf16 via-pred" $lib --model yolov6m --batch 8 --size 1280 --dtype bf16 --via-pred
done
cat $out
