"""Row-f2/f3 kernels in isolation at headline-like sizes: LayerNorm / BatchNorm (fwd, bwd), depthwise 7x7 (fwd, dgrad,
wgrad) on a (16, 128, 128, 48) backbone activation, MS-DSSIM on (64, 512, 512, 1).  Prints ms, TFLOP/s, GB/s from the
library's own event profiler (algorithmic bytes / flops as declared at the launch sites)."""
import ctypes, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
lib = L.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(0)
D = lambda *s: DeviceArray.from_numpy(rng.standard_normal(s).astype(np.float32))
Z = lambda *s: DeviceArray.zeros(s)
N, H, W, C = 16, 128, 128, 48
npix = N * H * W
x, dy, y, dx = D(N, H, W, C), D(N, H, W, C), Z(N, H, W, C), Z(N, H, W, C)
gamma, beta, mm, mv = D(C), D(C), Z(C), DeviceArray.from_numpy(np.ones(C, np.float32))
dg, db, saved = Z(C), Z(C), Z(2 * C)
k, kb, dk, dkb = D(7, 7, C), D(C), Z(7, 7, C), Z(C)
B, HH = 64, 512
yt, yp, gp = D(B, HH, HH, 1), D(B, HH, HH, 1), Z(B, HH, HH, 1)
loss = Z(8)
L.check(lib.dl4ds_profile_enable(1))
for it in range(reps + 1):
    if it == 1:
        L.check(lib.dl4ds_profile_enable(1))          # drop the first (cold) pass
    L.check(lib.dl4ds_op_layernorm_fwd(x.ptr, gamma.ptr, beta.ptr, y.ptr, npix, C, 1e-3, 1))
    L.check(lib.dl4ds_op_layernorm_bwd(x.ptr, y.ptr, dy.ptr, gamma.ptr, dx.ptr, dg.ptr, db.ptr, npix, C, 1e-3, 1, 0))
    L.check(lib.dl4ds_op_batchnorm_fwd(x.ptr, gamma.ptr, beta.ptr, mm.ptr, mv.ptr, y.ptr, saved.ptr, npix, C, 1e-3, 0.99, 1, 1))
    L.check(lib.dl4ds_op_batchnorm_bwd(x.ptr, y.ptr, dy.ptr, gamma.ptr, saved.ptr, dx.ptr, dg.ptr, db.ptr, npix, C, 1, 0))
    L.check(lib.dl4ds_op_dwconv_fwd(x.ptr, k.ptr, kb.ptr, y.ptr, N, H, W, C, 7))
    L.check(lib.dl4ds_op_dwconv_bwd(x.ptr, k.ptr, dy.ptr, dx.ptr, dk.ptr, dkb.ptr, N, H, W, C, 7, 0))
    L.check(lib.dl4ds_op_loss(6, yt.ptr, yp.ptr, gp.ptr, B, HH, HH, 1, loss.ptr))
buf = ctypes.create_string_buffer(1 << 16)
L.check(lib.dl4ds_profile_report(buf, len(buf)))
for kk, v in json.loads(buf.value.decode()).items():
    print(f"{kk:28s} n={v['n']:3d} avg_ms={v['ms']/v['n']:8.4f} tflops={(v['flops']/(v['ms']*1e-3)/1e12) if v['flops'] else 0:7.2f} gbps={v['bytes']/(v['ms']*1e-3)/1e9:8.1f}")
