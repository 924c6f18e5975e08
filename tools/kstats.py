"""Compact view of a rocprofv3 *kernel_stats.csv: short kernel name, calls, average us, share."""
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r'\(anonymous namespace\)::', '', r['Name'])
    name = re.sub(r'^void ', '', name).split('(')[0]
    print(f"{name[:48]:48s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:10.1f} us {float(r['Percentage']):6.2f} %")
