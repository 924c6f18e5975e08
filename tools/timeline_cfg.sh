#!/bin/bash
# GPU box: kernel-trace of tools/bench_configs.py <cfg> <batch>: busy time vs wall span of the traced kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tlc
rocprofv3 --kernel-trace -d /tmp/tlc --output-format csv -- python $R/tools/bench_configs.py "$@" > /dev/null 2>&1
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob('/tmp/tlc/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam' in r[2]]
lo, hi = adam[-3] + 1, adam[-2] + 1          # one full step between two Adam launches
busy = sum(e - s for s, e, _ in rows[lo:hi])
span = rows[hi - 1][1] - rows[lo][0]
gaps = sorted((rows[i + 1][0] - rows[i][1]) for i in range(lo, hi - 1))
print(f'launches {hi - lo}  busy {busy / 1e6:.3f} ms  span {span / 1e6:.3f} ms  idle {100 * (1 - busy / span):.1f} %  '
      f'median gap {gaps[len(gaps) // 2] / 1e3:.2f} us')
PY
