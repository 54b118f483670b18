#!/bin/bash
# A/B of the pipelined 3x3 kernel against the best generic variants on the yololps layer shapes (B=32).
cd "$(dirname "$0")/../.."
run() { python tools/conv_bench.py --batch ${B:-32} --k 3 --cin $1 --cout $2 --hw $3 --variant $4 --iters 30 2>&1 | tail -1; }
for shape in "64 64 160" "128 128 80" "256 256 40" "512 512 20" "128 128 40" "64 64 80" "256 256 20"; do
  set -- $shape
  if [ "$1" = 64 ]; then gens="4,1 1,1 1,2"; pipes="33,3"; else gens="3,1 3,2 5,2 0,1"; pipes="32,3 34,3"; fi
  for v in $gens $pipes; do run $1 $2 $3 $v; done
done
