#!/bin/bash
# stride-2 3x3 layers of yololps (B=32) through the generic implicit-GEMM kernel, best variants
cd "$(dirname "$0")/../.."
run() { python tools/conv_bench.py --batch 32 --k 3 --s 2 --cin $1 --cout $2 --hw $3 --sl $4 --variant $5 --iters 20 2>&1 | tail -1; }
run 32 64 320 1 4,1; run 64 128 160 2 5,1; run 64 128 160 2 3,1; run 128 256 80 3 3,2; run 256 512 40 4 3,2; run 64 64 160 2 4,1; run 64 64 80 3 4,1; run 128 128 40 4 5,2
