#!/bin/bash
# Is a ring kernel's operand fetch bound by the chip (all CUs pulling together) or by each CU's own request latency?  The same layer on
# 256 / 128 / 64 persistent workgroups (LP_PIPE_MAXWG): chip-bound -> time grows less than the work per workgroup; latency-bound -> proportionally.
cd "$(dirname "$0")/../.."
for wg in 256 128 64; do
  echo "== LP_PIPE_MAXWG=$wg"
  LP_PIPE_MAXWG=$wg python tools/conv_bench.py --batch 32 --k 3 --s 2 --cin 128 --cout 256 --hw 80 --sl 3 --variant 48,3 --iters 30 2>&1 | tail -1
  LP_PIPE_MAXWG=$wg python tools/conv_bench.py --batch 32 --k 3 --s 1 --cin 256 --cout 256 --hw 40 --sl 4 --variant 42,3 --iters 30 2>&1 | tail -1
  LP_PIPE_MAXWG=$wg python tools/conv_bench.py --batch 32 --k 3 --s 1 --cin 128 --cout 128 --hw 80 --sl 3 --variant 42,3 --iters 30 2>&1 | tail -1
done
