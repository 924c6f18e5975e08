"""Print the per-tag breakdown of a bench.py JSON line (DL4DS_BENCH_BREAKDOWN=1), largest first."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
b = d['breakdown']
tot = sum(v['ms_per_step'] for v in b.values())
for k, v in sorted(b.items(), key=lambda kv: -kv[1]['ms_per_step'])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"{k:36s} n={v['launches_per_step']:6.1f} ms={v['ms_per_step']:8.3f} {100 * v['ms_per_step'] / tot:5.1f}%  "
          f"TF={(v['tflops'] or 0):6.1f} GB/s={(v['gbps'] or 0):7.0f}")
print('sum of kernels ms', round(tot, 3), ' step ms', round(d['ms_per_step'], 3))
