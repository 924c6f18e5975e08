"""HBM traffic per launch per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced
streaming reads -> doubled here; both counters are in KiB.  WRITE_SIZE is uncalibrated (taken as is)."""
import csv, glob, sys, collections, json, re


def collect(root, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    return agg


fetch = collect(sys.argv[1], 'FETCH_SIZE')
write = collect(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    m = re.search(r'(conv_\w+)<([^>]*)>', k)
    args = m.group(2).replace(' ', '').split(',') if m else []
    if m and 'igemm_db' in m.group(1):
        args = args[:5]                      # bench.py's tag carries <KS,MT,NT,WM,WN> only
    if m and m.group(1).startswith('conv_wgrad_rows'):
        args = ['3', args[0], '1', args[1]]  # kernel <CIT,WCO> -> bench.py's tag <KS,CIT,COT,WCO>
    elif m and m.group(1).startswith('conv_wgrad_kernel'):
        args = args[:4]                      # ... and <KS,CIT,COT,WCO> for wgrad (drop the prefetch flag)
    tag = (m.group(1).replace('_kernel', '').replace('rows_ws', 'rows') + '<' + ','.join(args) + '>') if m else k[:60]
    f = sum(fetch[k]) / len(fetch[k])
    w = sum(write.get(k, [0])) / max(len(write.get(k, [0])), 1)
    out[tag] = {'launches': len(fetch[k]), 'fetch_kib_raw': f, 'write_kib_raw': w,
                'hbm_bytes_per_launch': (2.0 * f + w) * 1024.0}
print(json.dumps(out, indent=1))
