#!/bin/bash
# round 4: LDS counters of the block-tiled kernel on two layers: 256->256 @40^2 (10 x 40 tiles: pixel blocks cross halo rows) and
# 128->128 @80^2 (5 x 80 tiles: no block crosses a row)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for spec in "256 256 40 5" "128 128 80 3"; do
  set -- $spec
  out=$GRAFT_REPO_ROOT/gpurun_out/pmc_l_$1_$3
  rm -rf $out
  cd /tmp && export TMPDIR=/tmp
  i=0
  for pmc in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 -M --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --iters 5 --batch 32 --k 3 --cin $1 --cout $2 --hw $3 --sl $4 --variant 42,3 > /dev/null 2>&1
  done
  cd "$GRAFT_REPO_ROOT"
  echo "== $1->$2 @$3"; python3 tools/micro/pmc_conv_summary.py $out
done > gpurun_out/r4_pmc_layers.txt 2>&1
cat gpurun_out/r4_pmc_layers.txt
