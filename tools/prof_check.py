"""Library profiler vs itself: a kernel timed alone, after a long kernel, and at several sizes (python tools/prof_check.py).
rocprofv3 --kernel-trace --stats of the same command gives the reference durations."""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
lib = L.lib()
rng = np.random.default_rng(0)
def rep():
    buf = ctypes.create_string_buffer(1 << 16)
    L.check(lib.dl4ds_profile_report(buf, len(buf)))
    return json.loads(buf.value.decode())
xc = DeviceArray.from_numpy(rng.standard_normal((16, 512, 512, 8)).astype(np.float32))
wc = DeviceArray.from_numpy((rng.standard_normal((3, 3, 8, 8)) * 0.1).astype(np.float32))
bc = DeviceArray.from_numpy(rng.standard_normal((8,)).astype(np.float32))
yc = DeviceArray.zeros((16, 512, 512, 8))
for H in (512, 256, 64):
    x = DeviceArray.from_numpy(rng.standard_normal((16, H, H, 8)).astype(np.float32))
    y = DeviceArray.zeros((16, H // 2, H // 2, 8))
    for mode in ('alone', 'after conv'):
        for it in range(2):
            L.check(lib.dl4ds_profile_enable(1))
            for _ in range(20):
                if mode == 'after conv':
                    L.check(lib.dl4ds_op_conv2d_fwd(xc.ptr, wc.ptr, bc.ptr, None, yc.ptr, 16, 512, 512, 8, 8, 3, 1, 0))
                L.check(lib.dl4ds_op_maxpool2_fwd(x.ptr, y.ptr, 16, H, H, 8))
            r = rep()
            L.check(lib.dl4ds_profile_enable(0))
        print(H, mode, {k: round(v['ms'] / v['n'] * 1e3, 1) for k, v in r.items()}, 'us per launch', flush=True)
