# Same-box A/B of two builds of the library: alternating runs of the default benchmark.
#   bash tools/micro/ab_bench.sh yolo-lp_amd/libyololp_hip_prev.so yolo-lp_amd/libyololp_hip.so
# (build the "previous" library from `git archive <rev> yolo-lp_amd/csrc include` in a scratch directory.)
for i in 1 2; do
  for lib in "$@"; do
    echo "$lib"
    LP_HIP_LIB=$PWD/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  done
done
