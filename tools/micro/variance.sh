for i in 1 2 3; do
  s=$(date +%s.%N)
  timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
  e=$(date +%s.%N); echo "wall $(echo "$e - $s" | bc) s"
done
