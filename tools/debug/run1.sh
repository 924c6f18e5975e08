mkdir -p gpurun_out/r1
for v in "THR=16" "THR=8" "THR=4" "SX=16" "SX=12" "SX=32"; do
  export DL4DS_TMP_WG_THR=16; unset DL4DS_TMP_WG_SX
  case $v in THR=*) export DL4DS_TMP_WG_THR=${v#THR=};; SX=*) export DL4DS_TMP_WG_SX=${v#SX=};; esac
  DL4DS_BENCH_BREAKDOWN=1 python bench.py --no-cpu-baseline --no-b16 --no-unfolded --batch 16 > gpurun_out/r1/b16_$v.json 2>/dev/null
  python - <<EOF
import json
d=json.loads(open('gpurun_out/r1/b16_$v.json').read().strip().splitlines()[-1])
B=d['breakdown']
w=sum(x['ms_per_step'] for k,x in B.items() if 'wgrad' in k or 'slab' in k)
print('$v', round(d['value'],1), 'wgrad-family ms', round(w,3), {k:round(x['ms_per_step'],3) for k,x in B.items() if k.startswith('conv_wino_wgrad') or k.startswith('conv_wgrad_rows<3,3') or k.startswith('wino_wgrad')})
EOF
done
