"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, mean of each counter per dispatch."""
import csv, glob, sys, collections
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if filt and filt not in k:
            continue
        agg[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f'   {c:36s} n={len(v):3d} mean={sum(v)/len(v):.4g}')
