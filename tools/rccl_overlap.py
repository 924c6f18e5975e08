"""Launch-ordered timeline of the LAST train step of a `DL4DS_FORCE_DIST=1 python bench.py ...` run from a rocprofv3
--kernel-trace CSV, showing where the RCCL all-reduce kernels (communication stream) sit relative to the backward-pass
kernels (compute stream) and to Adam, which waits for the last bucket.
Usage: rccl_overlap.py <rocprof output dir> [label]"""
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append(dict(s=int(r['Start_Timestamp']), e=int(r['End_Timestamp']), k=r['Kernel_Name'],
                         q=r.get('Queue_Id', '?'), st=r.get('Stream_Id', '?'),
                         g=int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0), wg=int(r.get('Workgroup_Size_X', 0) or 0)))
rows.sort(key=lambda r: r['s'])
is_nccl = lambda r: 'nccl' in r['k'].lower() or 'rccl' in r['k'].lower()      # (incl. standin_for_rccl_allreduce_kernel)
adam = [i for i, r in enumerate(rows) if 'adam' in r['k']]
# one step = everything after the previous step's last Adam up to and including this step's last Adam
n_adam_per_step = 2 if len(sys.argv) > 2 and 'cfg5' in sys.argv[2] else 1
lo = adam[-1 - n_adam_per_step] + 1 if len(adam) > n_adam_per_step else 0
hi = adam[-1] + 1
step = rows[lo:hi]
t0 = step[0]['s']


def short(k):
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = re.sub(r'^void ', '', k)
    return re.sub(r'\(.*$', '', k)[:64]


print(f'# {sys.argv[2] if len(sys.argv) > 2 else ""}: last train step, {len(step)} kernel launches; times relative to its first kernel')
print(f'# {"start":>10s} {"end":>10s} {"dur":>8s}  queue/stream  kernel            (* = RCCL kernel; "|| n" = compute kernels running concurrently)')
nccl = [r for r in step if is_nccl(r)]
comp = [r for r in step if not is_nccl(r)]
for r in step:
    mark = '*' if is_nccl(r) else ' '
    extra = ''
    if is_nccl(r):
        ov = [c for c in comp if c['s'] < r['e'] and c['e'] > r['s']]
        ovt = sum(min(c['e'], r['e']) - max(c['s'], r['s']) for c in ov)
        extra = f'   || {len(ov)} compute kernels, {ovt / 1e3:.1f} us of compute inside this collective\'s {((r["e"] - r["s"]) / 1e3):.1f} us'
    print(f'{mark} {(r["s"] - t0) / 1e3:9.1f}us {(r["e"] - t0) / 1e3:9.1f}us {(r["e"] - r["s"]) / 1e3:7.1f}us  q{r["q"]}/s{r["st"]}  {short(r["k"])}{extra}')
if nccl:
    last_adam = [r for r in step if 'adam' in r['k']]
    first_adam_start = min(r['s'] for r in last_adam)
    nccl_t = sum(r['e'] - r['s'] for r in nccl)
    hidden = 0
    for r in nccl:
        for c in comp:
            if 'adam' in c['k']:
                continue
            hidden += max(0, min(c['e'], r['e']) - max(c['s'], r['s']))
    tail = max(0, max(r['e'] for r in nccl) - max(c['e'] for c in comp if c['s'] < first_adam_start and 'adam' not in c['k']))
    print(f'# {len(nccl)} RCCL kernels, {nccl_t / 1e3:.1f} us in total; {hidden / 1e3:.1f} us of it under backward-pass kernels; '
          f'the last collective ends {tail / 1e3:.1f} us after the last backward kernel; Adam starts at '
          f'{(first_adam_start - t0) / 1e3:.1f} us (after the last collective: {first_adam_start >= max(r["e"] for r in nccl)})')
else:
    print('# no RCCL kernel in this trace')
