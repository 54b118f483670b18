#!/bin/bash
# Everything profiles/ holds for a round, generated on the GPU box in one go:  bash tools/make_profiles.sh r04
# (writes gpurun_out/prof_<tag>/; copy what is to be judged into profiles/ afterwards: tools/collect_profiles.sh <tag>)
set -o pipefail
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. the bench line + per-op table (plain run)
timeout -k 10 280 python3 $root/bench.py --steps 50 --warmup 5 --detail $out/per_op.txt > $out/bench_plain.json 2> $out/bench_plain.err || exit 1
# 2. the same command under rocprofv3 -M --kernel-trace --stats (six batches in flight)
timeout -k 10 400 rocprofv3 -M --kernel-trace --stats --output-format csv -d $out/kt -o kt -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/bench_rocprof.json 2> $out/bench_rocprof.err || exit 2
cp $(ls $out/kt/*kernel_stats.csv | head -1) $out/kernel_stats.csv
python3 $root/tools/micro/kstats.py $out/kt > $out/kernel_summary_inflight6.txt
# 2b. one batch in flight, one execution lane (kernels do not overlap: a dispatch's duration is its own): the kernels of a step in
#     issue order, and the roofline of the 3x3 layers from the trace (bench.py reports it as roofline.frac when the hash matches)
timeout -k 10 400 rocprofv3 -M --kernel-trace --stats --output-format csv -d $out/kt1 -o kt -- python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --inflight 1 --single-lane 1 > $out/bench_rocprof_inflight1.json 2> $out/bench_rocprof_inflight1.err || exit 2
cp $(ls $out/kt1/*kernel_stats.csv | head -1) $out/kernel_stats_inflight1.csv
python3 $root/tools/micro/kstats.py $out/kt1 > $out/kernel_summary_inflight1.txt
python3 $root/tools/micro/step_kernels.py $out/kt1 50 > $out/step_kernels_inflight1.txt
python3 $root/tools/roofline_from_trace.py $out/kt1 $out/bench_rocprof_inflight1.json --steps 50 > $out/roofline.json || exit 2
# 3. HBM traffic of the 3x3 launches: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), one batch in flight
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 -M --kernel-trace --pmc $c --output-format csv -d $out/pmc_$c -o p -- python3 $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --single-lane 1 > /dev/null 2> $out/pmc_$c.err || exit 3
done
python3 $root/tools/pmc_traffic.py $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE --steps 3 --bench $out/bench_plain.json > $out/pmc_traffic.json || exit 4
# 4. counters of the pipelined kernel on the 256 -> 256 @ 40x40 layer
bash $root/tools/micro/pmc_conv.sh $out/pmc_pd3 --batch 32 --k 3 --cin 256 --cout 256 --hw 40 --variant 32,3 > /dev/null 2>&1
python3 $root/tools/micro/pmc_conv_summary.py $out/pmc_pd3 > $out/pmc_conv_256x256_40_Pd3.txt
# ... and of the block-tiled 16x16x32 kernel that runs this layer since round 4 (LP_VARIANT_PIPE16_V0)
bash $root/tools/micro/pmc_conv.sh $out/pmc_v0 --batch 32 --k 3 --cin 256 --cout 256 --hw 40 --variant 42,3 > /dev/null 2>&1
python3 $root/tools/micro/pmc_conv_summary.py $out/pmc_v0 > $out/pmc_conv_256x256_40_V0.txt
# 5. the other single-GPU configurations of BASELINE.json: bench lines, per-op tables, kernel-trace summaries
sec() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', 'value', d['value'], 'value_inflight1', d['value_inflight1'], 'ms_per_step', d['ms_per_step'], 'roofline', r['bound'], r['achieved'], r['unit'], 'frac', r['frac'], 'forward_device_ms', r['forward_device_ms'], 'nms_device_ms', r['nms_device_ms'], 'step_ms_inflight1', d['step_ms_inflight1'])"; }
: > $out/secondary_configs.txt
timeout -k 10 300 python3 $root/bench.py --model yololpn --batch 128 --steps 30 --warmup 3 --no-cpu-baseline --detail $out/per_op_yololpn_bs128.txt 2>/dev/null | tee $out/bench_yololpn_bs128.json | sec "yololpn 640 bs128 f16 (detections-only forward)" >> $out/secondary_configs.txt
timeout -k 10 300 python3 $root/bench.py --model yololpn --batch 128 --steps 30 --warmup 12 --no-cpu-baseline --via-pred 2>/dev/null | sec "yololpn 640 bs128 f16 (--via-pred)" >> $out/secondary_configs.txt
timeout -k 10 400 python3 $root/bench.py --model yolov6m --batch 8 --size 1280 --dtype bf16 --steps 20 --warmup 3 --no-cpu-baseline --detail $out/per_op_yolov6m_1280.txt 2>/dev/null | tee $out/bench_yolov6m_1280.json | sec "yolov6m 1280 bs8 bf16 (detections-only forward)" >> $out/secondary_configs.txt
timeout -k 10 300 python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --via-pred 2>/dev/null | sec "yololps 640 bs32 f16 (--via-pred)" >> $out/secondary_configs.txt
timeout -k 10 300 python3 $root/bench.py --steps 50 --warmup 5 --no-cpu-baseline --single-lane 0 2>/dev/null | sec "yololps 640 bs32 f16 (three execution lanes per forward at six in flight)" >> $out/secondary_configs.txt
rm -rf $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
echo done; ls $out
