#!/bin/bash
# Copies what is to be judged from gpurun_out/prof_<tag>/ (tools/make_profiles.sh) into profiles/<tag>_*
tag=${1:-r04}; src=gpurun_out/prof_$tag; dst=profiles
for f in bench_plain.json bench_rocprof.json bench_rocprof_inflight1.json per_op.txt kernel_stats.csv kernel_stats_inflight1.csv kernel_summary_inflight1.txt \
         kernel_summary_inflight6.txt step_kernels_inflight1.txt roofline.json pmc_traffic.json pmc_conv_256x256_40_Pd3.txt pmc_conv_256x256_40_V0.txt secondary_configs.txt \
         per_op_yololpn_bs128.txt per_op_yolov6m_1280.txt bench_yololpn_bs128.json bench_yolov6m_1280.json \
         kernel_summary_yolov6m_1280_bs8_bf16.txt roofline_yolov6m_1280_bs8_bf16.json pmc_traffic_yolov6m_1280_bs8_bf16.json bench_rocprof_yolov6m_1280_bs8_bf16.json \
         kernel_summary_yololpn_640_bs128_f16.txt roofline_yololpn_640_bs128_f16.json pmc_traffic_yololpn_640_bs128_f16.json bench_rocprof_yololpn_640_bs128_f16.json; do
  [ -f $src/$f ] && cp $src/$f $dst/${tag}_$f
done
ls $dst/${tag}_*
