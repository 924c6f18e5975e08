#!/bin/bash
# GPU box: where a step's idle time sits.  tools/step_gaps.sh cfg5 -> gpurun_out/step_gaps_<cfg>.txt: busy / span / idle of one step between
# two Adam launches of `bench.py --config <cfg>` under rocprofv3 --kernel-trace, the histogram of gaps and the largest ones with neighbours.
CFG=${1:-cfg5}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sg_$CFG
rocprofv3 --kernel-trace -d /tmp/sg_$CFG --output-format csv -- python $R/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-unfolded --no-b16 --no-profile > /dev/null 2>&1
python - $CFG > $R/gpurun_out/step_gaps_$CFG.txt <<'PY'
import csv, glob, sys, collections
cfg = sys.argv[1]
rows = []
for f in glob.glob(f'/tmp/sg_{cfg}/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam' in r[2]]
per = 2 if cfg == 'cfg5' else 1                   # Adam launches per step
lo, hi = adam[-1 - 2 * per] + 1, adam[-1 - per] + 1
busy = sum(e - s for s, e, _ in rows[lo:hi])
span = rows[hi - 1][1] - rows[lo][0]
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2][:48], rows[i + 1][2][:48]) for i in range(lo, hi - 1)]
print(f'launches {hi - lo}  busy {busy / 1e6:.3f} ms  span {span / 1e6:.3f} ms  idle {(span - busy) / 1e6:.3f} ms = {100 * (1 - busy / span):.1f} %')
h = collections.Counter(min(int(g / 1e3), 50) for g, _, _ in gaps)
print('gap histogram (us: count):', sorted(h.items()))
print('sum of gaps > 10 us: %.3f ms' % (sum(g for g, _, _ in gaps if g > 1e4) / 1e6))
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print(f'{g / 1e3:8.1f} us  after {a:48s} before {b}')
PY
