set -e
run() { timeout -k 10 120 python tools/conv_bench.py --batch 32 "$@" 2>&1 | grep variant; }
run --k 1 --cin 64 --cout 64 --hw 160 --variant 5,2
run --k 1 --cin 64 --cout 64 --hw 160 --variant 5,3
run --k 1 --cin 64 --cout 64 --hw 160 --variant 1,2
run --k 1 --cin 128 --cout 128 --hw 80 --variant 3,1
run --k 1 --cin 128 --cout 128 --hw 80 --variant 6,2
run --k 3 --cin 64 --cout 64 --hw 160 --variant 1,2
run --k 3 --cin 256 --cout 256 --hw 40 --variant 3,1
