# GPU box: cfg2 at per-GPU batch 8 / 16 / 64 with conv_split strips of 8 / 16 / 32 rows (DL4DS_SPLIT_R, a test hook) -> gpurun_out/rs/ (round 6: 16 rows only where 32 leave CUs idle)
mkdir -p gpurun_out/rs
export DL4DS_TEST_HOOKS=1
for B in 8 16 64; do for R in 8 16 32; do
  DL4DS_SPLIT_R=$R DL4DS_BENCH_BREAKDOWN=1 python bench.py --batch $B --no-cpu-baseline --no-unfolded --no-b16 > gpurun_out/rs/b${B}_r$R.json 2> gpurun_out/rs/b${B}_r$R.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/rs/*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); b=j['breakdown']
        print(f, round(j['value'],1), round(j['ms_per_step'],3), {k:round(v['ms_per_step'],3) for k,v in b.items() if k.startswith(('conv_split','conv_wino<3,3>'))})
    except Exception as e: print(f,'ERR',e)
PY
