"""Pretty-print a bench.py JSON line (with DL4DS_BENCH_BREAKDOWN=1)."""
import json, sys
d = json.load(open(sys.argv[1]))
print('value %.1f %s  ms/step %.3f  cpu %s  gpu/cpu %s' % (d['value'], d['unit'], d['ms_per_step'], d.get('cpu_baseline'), d.get('gpu_over_cpu')))
print('roofline', d['roofline'])
tot = 0
for k, v in (d.get('breakdown') or {}).items():
    tot += v['ms_per_step']
    print(f"{k:28s} n={v['launches_per_step']:5.1f} ms={v['ms_per_step']:7.3f} tf={(v['tflops'] or 0):7.2f} gbps={v['gbps']:8.1f}")
print('sum of kernels ms/step', tot)
