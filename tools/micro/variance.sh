# Run-to-run spread of the default benchmark on one box: images/s, ms per step, 3x3 TFLOP/s of three runs.
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
