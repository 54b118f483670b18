#!/bin/bash
# round 4: are the LDS "bank conflict" cycles of the pipelined kernels read-read conflicts or LDS-DMA writes meeting reads?  Counters of the
# 32x32x16 kernel (PIPE_D, 256->256 @40^2) in the product build and in the ablation build without DMA in the loop (make abl ABL=1)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for lib in base abl1; do
  if [ $lib = base ]; then unset LP_HIP_LIB; else export LP_HIP_LIB=$GRAFT_REPO_ROOT/yolo-lp_amd/libyololp_hip_$lib.so; fi
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_abl_$lib
  rm -rf $out
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 200 rocprofv3 -M --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $out/p1 -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --iters 5 --batch 32 --k 3 --cin 256 --cout 256 --hw 40 --variant 32,3 > /dev/null 2>&1
  cd "$GRAFT_REPO_ROOT"
  echo "== $lib"; python3 tools/micro/pmc_conv_summary.py $out
done > gpurun_out/r4_pmc_abl.txt 2>&1
cat gpurun_out/r4_pmc_abl.txt
