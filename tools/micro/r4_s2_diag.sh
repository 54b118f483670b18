#!/bin/bash
# Stride-2 3x3 layers of yololps (B=32) through conv_mfma_kernel<KS=3,S=2>: time of every tile variant, then the L2 -> fabric fetch
# bytes (FETCH_SIZE) and L2 hit / miss counts of the variant the network uses.  Question: is a stride-2 halo (4x the pixels of a
# stride-1 one) re-fetched from the Infinity Cache once per K-chunk?
# usage: bash tools/micro/r4_s2_diag.sh <out_dir>
out=$1; mkdir -p $out
root=$(cd "$(dirname "$0")/../.." && pwd)
cd $root
run() { python tools/conv_bench.py --batch 32 --k 3 --s 2 --cin $1 --cout $2 --hw $3 --sl $4 --variant $5 --iters 20 2>&1 | tail -1; }
{
for v in 0,2 1,2 3,2 3,1 5,2 5,1 4,2; do run 128 256 80 3 $v; done
for v in 0,2 3,2 5,2 5,1; do run 256 512 40 4 $v; done
for v in 3,1 3,2 5,1 5,2 1,2; do run 64 128 160 2 $v; done
} > $out/s2_variants.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for layer in "128 256 80 3 3,2" "256 512 40 4 3,2" "64 128 160 2 5,1"; do
  set -- $layer
  tag=${1}x${2}_$3
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    timeout -k 10 200 rocprofv3 -M --pmc $c --kernel-trace --output-format csv -d $out/pmc_${tag}_$n -o p -- python3 $root/tools/conv_bench.py --iters 5 --batch 32 --k 3 --s 2 --cin $1 --cout $2 --hw $3 --sl $4 --variant $5 > /dev/null 2>&1 || exit 3
  done
done
python3 - $out <<'P'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + '/pmc_*')):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'conv_mfma' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d.split('/')[-1], {k: (sum(v) / len(v), len(v)) for k, v in acc.items()})
P
