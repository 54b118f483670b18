#!/bin/bash
# Kernel-trace summary of the post-processing kernels (score / sort / greedy) for the secondary configurations.
#   bash tools/micro/nms_trace.sh <tag>        -> gpurun_out/nms_trace_<tag>.txt
set -e
tag=${1:-x}
out=gpurun_out/nms_trace_$tag.txt
mkdir -p gpurun_out
cd /tmp 2>/dev/null || true
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
: > $out
run() {
    name=$1; shift
    rm -rf gpurun_out/kt_nms
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_nms -o kt -- python3 bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --inflight 1 > gpurun_out/kt_nms.json 2> gpurun_out/kt_nms.err
    echo "== $name" >> $out
    python3 tools/micro/kstats.py gpurun_out/kt_nms | grep -i "score_kernel\|sort_kernel\|greedy_kernel\|head_cls\|sort_\|nms" >> $out || true
}
run "yololps bs32 f16 (via-pred)" --via-pred
run "yololps bs32 f16 (det)"
run "yololpn bs128 f16 (via-pred)" --model yololpn --batch 128 --via-pred
run "yolov6m 1280 bs8 bf16 (via-pred)" --model yolov6m --batch 8 --size 1280 --dtype bf16 --via-pred
cat $out
