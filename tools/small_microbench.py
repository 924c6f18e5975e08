"""8->8 3x3 conv at 512^2, batch 16 (the HBM-bound tail layers of cfg2): fwd / wgrad in isolation (MB_N / MB_H / MB_W / MB_CI / MB_CO
override the shape: MB_CI=16 MB_CO=16 is the U-Net's 16-channel level; MB_RELU=0|1, MB_FWD_ONLY=1; short runs see the clock ramp:
use >= 100 repetitions for rates)."""
import ctypes, json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
lib = L.lib()
N, H, W, CI, CO = (int(os.environ.get(k, d)) for k, d in (('MB_N', 16), ('MB_H', 512), ('MB_W', 512), ('MB_CI', 8), ('MB_CO', 8)))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
relu = int(os.environ.get('MB_RELU', 1))
fwd_only = bool(os.environ.get('MB_FWD_ONLY'))
rng = np.random.default_rng(0)
x = DeviceArray.from_numpy(rng.standard_normal((N, H, W, CI)).astype(np.float32))
w = DeviceArray.from_numpy((rng.standard_normal((3, 3, CI, CO)) * 0.1).astype(np.float32))
b = DeviceArray.from_numpy(rng.standard_normal((CO,)).astype(np.float32))
y = DeviceArray.zeros((N, H, W, CO))
dz = DeviceArray.from_numpy(rng.standard_normal((N, H, W, CO)).astype(np.float32))
dx = DeviceArray.zeros((N, H, W, CI))
dw = DeviceArray.zeros((3, 3, CI, CO))
L.check(lib.dl4ds_profile_enable(1))
for _ in range(reps):
    L.check(lib.dl4ds_op_conv2d_fwd(x.ptr, w.ptr, b.ptr, None, y.ptr, N, H, W, CI, CO, 3, relu, 0))
    if not fwd_only:
        L.check(lib.dl4ds_op_conv2d_wgrad(x.ptr, dz.ptr, dw.ptr, N, H, W, CI, CO, 3, 0, 0))
buf = ctypes.create_string_buffer(1 << 16)
L.check(lib.dl4ds_profile_report(buf, len(buf)))
for k, v in json.loads(buf.value.decode()).items():
    print(f"{k:28s} n={v['n']:3d} avg_ms={v['ms']/v['n']:8.4f} tflops={(v['flops']/(v['ms']*1e-3)/1e12) if v['flops'] else 0:7.2f} gbps={v['bytes']/(v['ms']*1e-3)/1e9:8.1f}")
