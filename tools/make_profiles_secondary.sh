#!/bin/bash
# The trace roofline and the PMC traffic of the two other single-GPU configurations of BASELINE.json, as tools/make_profiles.sh takes them for
# the default one:  bash tools/make_profiles_secondary.sh r04   (writes gpurun_out/prof_<tag>/; tools/collect_profiles.sh copies into profiles/)
set -o pipefail
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
out=$root/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# files bench.py picks up by workload name (profiles/<tag>_roofline_<model>_<size>_bs<batch>_<dtype>.json, ..._pmc_traffic_...)
sec_cfg() {   # $1 = name, rest = bench.py arguments of the configuration
  name=$1; shift
  timeout -k 10 400 rocprofv3 -M --kernel-trace --output-format csv -d $out/kt_$name -o kt -- python3 $root/bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --inflight 1 --single-lane 1 > $out/bench_rocprof_$name.json 2> $out/kt_$name.err || return 2
  python3 $root/tools/micro/kstats.py $out/kt_$name > $out/kernel_summary_$name.txt
  python3 $root/tools/roofline_from_trace.py $out/kt_$name $out/bench_rocprof_$name.json --steps 10 > $out/roofline_$name.json || return 2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 500 rocprofv3 -M --kernel-trace --pmc $c --output-format csv -d $out/pmc_${c}_$name -o p -- python3 $root/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --inflight 1 --single-lane 1 > /dev/null 2> $out/pmc_${c}_$name.err || return 3
  done
  python3 $root/tools/pmc_traffic.py $out/pmc_FETCH_SIZE_$name $out/pmc_WRITE_SIZE_$name --steps 3 --bench $out/bench_rocprof_$name.json > $out/pmc_traffic_$name.json || return 4
  rm -rf $out/kt_$name $out/pmc_FETCH_SIZE_$name $out/pmc_WRITE_SIZE_$name     # (the raw traces are large: the summaries stay)
}
sec_cfg yolov6m_1280_bs8_bf16 --model yolov6m --batch 8 --size 1280 --dtype bf16 || echo "yolov6m profile set failed: $?"
sec_cfg yololpn_640_bs128_f16 --model yololpn --batch 128 || echo "yololpn profile set failed: $?"
echo done; ls $out
