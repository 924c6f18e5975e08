# tools/debug/ab.sh <git-ref-or-path-of-other-.so> : alternating bench runs of the tree's library and another build of it (DL4DS_HIP_LIB)
export DL4DS_BENCH_BREAKDOWN=1
P='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=j["breakdown"]; print(sys.argv[1], j["config"]["per_gpu_batch"], round(j["value"],1), round(j["ms_per_step"],4), j["config"]["loss_after_run"], {k: round(b[k]["ms_per_step"],4) for k in sys.argv[2:] if k in b})'
for rep in 1 2 3; do for B in 16 64; do
  python bench.py --batch $B --no-cpu-baseline --no-unfolded --no-b16 | python -c "$P" new "${@:2}"
  DL4DS_HIP_LIB=$1 python bench.py --batch $B --no-cpu-baseline --no-unfolded --no-b16 | python -c "$P" old "${@:2}"
done; done
