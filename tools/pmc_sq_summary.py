"""Per-kernel averages of the SQ counters collected by tools/pmc_sq.sh (one column per counter)."""
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
names = []
for root in sys.argv[1:]:
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
            k = re.sub(r'^void ', '', k)
            k = re.sub(r'\(.*$', '', k) + ' g=' + r.get('Grid_Size', '?')
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] not in names:
                names.append(r['Counter_Name'])
print('kernel'.ljust(64), ' '.join(n[-18:].rjust(18) for n in names))
rows = []
for k, d in agg.items():
    rows.append((sum(d.get('SQ_BUSY_CU_CYCLES', [0])), k, d))
for _, k, d in sorted(rows, key=lambda t: -t[0])[:24]:
    print(k[:64].ljust(64), ' '.join((f'{sum(d[n]) / len(d[n]):18.4g}' if n in d else ' ' * 18) for n in names))
