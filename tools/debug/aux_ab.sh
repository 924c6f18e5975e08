# GPU box: DL4DS_AUX_STREAM=1 against the default, alternating, cfg2 (B = 16, 64), cfg4, cfg5 (round 6: cfg2 +-0.1 %, cfg5 -1 %, cfg4 -18 %)
P='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["config"]["per_gpu_batch"], round(j["value"],1), round(j["ms_per_step"],4), j["roofline"]["kernel"], round(j["roofline"]["frac"],3))'
for rep in 1 2; do for C in cfg2:16 cfg2:64 cfg4:16 cfg5:8; do
  c=${C%%:*}; B=${C##*:}
  python bench.py --config $c --batch $B --no-cpu-baseline --no-unfolded --no-b16 | python -c "$P" "$c default"
  DL4DS_AUX_STREAM=1 python bench.py --config $c --batch $B --no-cpu-baseline --no-unfolded --no-b16 | python -c "$P" "$c aux"
done; done
