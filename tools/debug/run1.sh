for i in 1 2 3; do
  DL4DS_SPLIT=1 python bench.py --no-cpu-baseline --no-b16 --no-unfolded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split', round(d['value'],1), round(d['steady_state']['value'],1))"
  python bench.py --no-cpu-baseline --no-b16 --no-unfolded 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('wino ', round(d['value'],1), round(d['steady_state']['value'],1))"
done
