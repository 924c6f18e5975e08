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
def tag_of(k):
    if 'rec_tail_fwd' in k:
        return 'rec_tail_fwd'                # plain and LDS-staged form share bench.py's tag
    if 'rec_tail_bwd' in k:
        return 'rec_tail_bwd'
    if 'conv_split_kernel' in k:
        return 'conv_split<3,3>'             # (the compiled epilogue-operand form is not part of bench.py's tag)
    m = re.search(r'(conv\w+)<([^>]*)>', k)
    if not m:
        return k[:60]
    args = m.group(2).replace(' ', '').split(',')
    name = m.group(1).replace('conv_wino2_kernel', 'conv_wino_kernel').replace('conv_wino_wgrad2_kernel', 'conv_wino_wgrad_kernel')   # (round 4: second forms, same tags)
    if name.startswith('conv_narrow_pair_ws') or name.startswith('conv_narrow16_ws'):
        args = args[:1]                      # <NR> (the compiled epilogue form is not part of bench.py's tag)
    name = name.replace('conv_direct2', 'conv_direct')      # both generations of the stencil kernels share bench.py's tag
    if name.startswith('conv_wino_kernel'):
        args = args[:2]                      # <KQ,NT> (the epilogue form is not part of bench.py's tag)
    if name.startswith('conv_point'):
        return 'conv_point'                  # bench.py's tag has no template arguments: both tile variants merge
    if name.startswith('convlstm_seq'):
        args = args[:2]                      # <KS,F> (the tile-rows parameter is not part of bench.py's tag)
    if 'igemm_db' in name:
        args = args[:5]                      # bench.py's tag carries <KS,MT,NT,WM,WN> only
    if name.startswith('conv_wgrad_rows'):
        args = [args[2] if len(args) > 2 else '3', args[0], '1', args[1]]   # kernel <CIT,WCO,KS> -> bench.py's tag <KS,CIT,COT,WCO>
    elif name.startswith('conv_wgrad_kernel'):
        args = args[:4]                      # ... and <KS,CIT,COT,WCO> for wgrad (drop the prefetch flag)
    elif name.startswith('conv_stream_ws'):
        args = args[:4]                      # <KS,E,NT,MT> (drop the narrow-group flag: both forms share bench.py's tag)
    return name.replace('_kernel', '').replace('rows_ws', 'rows') + '<' + ','.join(args) + '>'


merged = {}
for k in fetch:
    t = merged.setdefault(tag_of(k), [[], []])
    t[0] += list(fetch[k])
    t[1] += list(write.get(k, []))
out = {}
for tag, (fl, wl) in merged.items():
    f = sum(fl) / len(fl)
    w = sum(wl) / max(len(wl), 1)
    out[tag] = {'launches': len(fl), 'fetch_kib_raw': f, 'write_kib_raw': w,
                'hbm_bytes_per_launch': (2.0 * f + w) * 1024.0}
print(json.dumps(out, indent=1))
