mkdir -p gpurun_out/r1
python bench.py > gpurun_out/r1/default.json 2> gpurun_out/r1/default.err
python - <<'EOF'
import json
d=json.loads(open('gpurun_out/r1/default.json').read().strip().splitlines()[-1])
print(round(d['value'],1), d['gpu_over_cpu'], d['roofline']['traffic'])
c=d['cpu_baseline']; print({k:c[k] for k in c if k not in ('sample',)})
print(c['sample'])
EOF
