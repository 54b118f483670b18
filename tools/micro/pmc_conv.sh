# PMC passes over one conv layer (tools/conv_bench.py): MFMA / LDS / VMEM busy counters of the conv kernel.
# usage: bash tools/micro/pmc_conv.sh <out_dir> <conv_bench args...>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_ANY SQ_INSTS_VMEM" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum" \
           "TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $out/p$i -o p -- python $GRAFT_REPO_ROOT/tools/conv_bench.py --iters 5 "$@" > /dev/null 2>&1
done
