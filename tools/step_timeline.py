"""One train step as a launch-ordered timeline from a rocprofv3 --kernel-trace CSV: the launches between the last
two adam kernels, with grid size and duration.  Usage: step_timeline.py <rocprof output dir> [adam launches per step]
(cfg5's CGAN step has two: discriminator and generator)"""
import csv, glob, sys, re
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0), int(r.get('Workgroup_Size_X', 0) or 0)))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam' in r[2]]
aps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = (adam[-1 - aps] + 1, adam[-1] + 1) if len(adam) > aps else (0, len(rows))
t0 = rows[lo][0]
tot = 0
for s, e, k, g, wg in rows[lo:hi]:
    name = re.sub(r'\(anonymous namespace\)::', '', k)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    tot += e - s
    print(f'{(s - t0) / 1e3:10.1f}us {(e - s) / 1e3:9.1f}us blocks={g // max(wg, 1):7d} {name[:70]}')
print(f'sum of kernel durations {tot / 1e6:.3f} ms, span {(rows[hi - 1][1] - t0) / 1e6:.3f} ms')
