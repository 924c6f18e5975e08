"""Condense rocprofv3 --kernel-trace --stats output: per kernel name, calls / total / average duration."""
import csv, glob, sys, collections, re
root = sys.argv[1]
rows = collections.OrderedDict()
files = glob.glob(root + '/**/*kernel_stats.csv', recursive=True)
if files:
    for r in csv.DictReader(open(files[0])):
        rows[r['Name']] = (int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs']), float(r['Percentage']))
else:   # fall back to the raw trace
    agg = collections.defaultdict(list)
    for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r['Kernel_Name']].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    tot = sum(sum(v) for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        rows[k] = (len(v), sum(v), sum(v) / len(v), 100 * sum(v) / tot)
print(f'{"kernel":90s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"pct":>6s}')
for k, (n, t, a, p) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    name = re.sub(r'\(anonymous namespace\)::', '', k)
    name = re.sub(r'^void ', '', name)
    print(f'{name[:90]:90s} {n:7d} {t/1e6:10.3f} {a/1e3:10.2f} {p:6.2f}')
