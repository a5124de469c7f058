for v in 21 20 22 23 40 41 42 43 11 13; do
  echo "variant $v: $(RN_K3_VARIANT=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readlines()[-1]); print(d["value"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"])')"
done
