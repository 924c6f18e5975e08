"""Phase sums of conv_stream_ws_kernel workgroups from a -DSTREAM_TRACE build (tools/variant_build.sh trace conv_stream.hip
-DSTREAM_TRACE; run anything with DL4DS_HIP_LIB=gpurun_variants/libdl4ds_trace.so; the first launches dump
gpurun_out/ws_trace_<k>.bin).   python tools/ws_trace.py gpurun_out/ws_trace_2.bin [mfmas per K loop]"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.float64)
mf = float(sys.argv[2]) if len(sys.argv) > 2 else 1296.0
steps = a[:, 5]
ok = steps > 0
a, steps = a[ok], steps[ok]
us = lambda col: a[:, col] / 100.0
print(f'{len(a)} workgroups with work, {steps.mean():.1f} steps each; per workgroup, mean (us):')
print(f'  MFMA waves : lifetime {us(4).mean():8.1f} | K loops {us(1).mean():8.1f} | waiting at X {us(2).mean():7.1f} | hand-over {us(3).mean():7.1f}')
print(f'  helpers    : lifetime {us(12).mean():8.1f} | drain {us(8).mean():8.1f} | helper barrier {us(9).mean():6.1f} | stage {us(10).mean():7.1f} | waiting at X/Y {us(11).mean():7.1f}')
print(f'  stage split: address + issue of the loads {us(14).mean():7.1f} | waiting for the data {us(15).mean():7.1f} | LDS writes (rest) {us(10).mean():7.1f}')
cyc = a[:, 0] / steps
print(f'  K loop: {cyc.mean():.0f} shader cycles per step = {mf * 32 / cyc.mean():.3f} of the matrix pipe ({mf:.0f} MFMAs x 32 cycles);'
      f' {us(1).sum() / steps.sum():.2f} us per step -> shader clock {cyc.mean() / (us(1).sum() / steps.sum()) :.0f} MHz')
simd = a[:, 13].astype(np.int64)
pl = np.stack([(simd >> (2 * w)) & 3 for w in range(8)], 1)
print('  SIMD of waves 0-7 (first workgroups):', pl[:4].tolist(), ' MFMA waves on distinct SIMDs in', int((np.sort(pl[:, :4], 1) == np.arange(4)).all(1).sum()), 'of', len(pl))
