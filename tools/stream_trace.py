"""Phase timeline of conv_stream_kernel workgroups from a -DSTREAM_TRACE build (tools/variant_build.sh trace conv_stream.hip
-DSTREAM_TRACE; run anything with DL4DS_HIP_LIB=gpurun_variants/libdl4ds_trace.so; the first launches of the <.,.,3,8>
variant dump gpurun_out/stream_trace_<k>.bin).   python tools/stream_trace.py gpurun_out/stream_trace_2.bin"""
import sys
from collections import defaultdict
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
t = a[:, :7].astype(np.int64)
hw = a[:, 7]
nst = int((t[0] > 0).sum())                      # stamps per workgroup: start, (staged, mfma done) x chunks, end
t = (t[:, :nst] - t[:, 0].min()) / 100.0         # us (s_memrealtime = 100 MHz)
cu, sh, se, tg = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7, (hw >> 16) & 0xf
d = np.diff(t, axis=1)
names = []
for c in range((nst - 2) // 2):
    names += [f'stage{c}', f'mfma{c}']
names += ['epilogue']
print(f'{len(a)} workgroups, kernel span {t[:, -1].max():.1f} us; phase durations in us: mean / p10 / p90')
for i, nme in enumerate(names):
    print(f'  {nme:9s} {d[:, i].mean():7.2f} {np.percentile(d[:, i], 10):7.2f} {np.percentile(d[:, i], 90):7.2f}')
print(f'  workgroup lifetime {(t[:, -1] - t[:, 0]).mean():.2f}')
xcd = np.arange(len(a)) % 8
groups = defaultdict(list)
for i, k in enumerate(zip(xcd.tolist(), se.tolist(), sh.tolist(), cu.tolist())):
    groups[k].append(i)
grid = np.linspace(0, t[:, -1].max(), 20000)
fr = np.zeros(3)
for k, idx in list(groups.items())[:16]:
    cnt = np.zeros_like(grid)
    for i in idx:
        for c in range((nst - 2) // 2):
            cnt += (grid >= t[i, 1 + 2 * c]) & (grid < t[i, 2 + 2 * c])
    fr += np.array([(cnt == j).mean() for j in range(3)])
print('share of the kernel time a CU has 0 / 1 / 2 workgroups in their MFMA phase (16 CUs):', (fr / 16).round(3))
k0 = sorted(groups)[3]
print('one CU', k0, ':')
for i in sorted(groups[k0], key=lambda i: t[i, 0])[:8]:
    print(f'  wg {i:5d} slot {tg[i]}  ', ' '.join(f'{x:8.2f}' for x in t[i]))
