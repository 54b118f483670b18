#!/bin/bash
for d in 0 1 5 2 8 16 10 26; do
  LP_DET_DBG=$d bash tools/micro/r3_kt1.sh r3_detabl/d$d > gpurun_out/r3_detabl_$d.log 2>&1
  echo "dbg=$d: $(grep -E 'head_det' gpurun_out/r3_detabl_$d.log | awk '{printf "%s us  ", $3}')"
done
