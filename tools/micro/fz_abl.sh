#!/bin/bash
# timing-only ablation builds of the fused stem kernel (make opt OPTS=1 OFLAGS=-DLP_FZ_ABL=k -> libyololp_hip_fzablK.so): op 2 of bench's per-op table
cd "$GRAFT_REPO_ROOT"
out=gpurun_out/fz_abl.txt
: > $out
for lib in yolo-lp_amd/libyololp_hip.so yolo-lp_amd/libyololp_hip_fzabl*.so; do
    LP_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --warmup 2 --detail gpurun_out/fz_abl_ops.txt > /dev/null 2> gpurun_out/fz_abl.err || { echo "$lib failed" >> $out; continue; }
    echo "$(basename $lib): $(sed -n 3p gpurun_out/fz_abl_ops.txt)" >> $out
done
cat $out
