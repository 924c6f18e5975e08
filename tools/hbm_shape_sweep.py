"""The narrow-channel HBM kernels at a fixed pixel count (16.8 M) but different image shapes: does the rate depend on the row
pitch (512 px x 32 B = 16 KB at the BASELINE sizes)?  python tools/hbm_shape_sweep.py [reps]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
lib = L.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rng = np.random.default_rng(0)
SHAPES = [(64, 512, 512), (16, 1024, 1024), (256, 256, 256)]
LAYERS = [tuple(int(v) for v in l.split("x")) for l in os.environ.get("SWEEP_LAYERS", "8x1,8x8,1x8,1x1").split(",")]
for CI, CO in LAYERS:
    for B, H, W in SHAPES:
        x = DeviceArray.from_numpy(rng.standard_normal((B * H * W * CI,)).astype(np.float32))
        w = DeviceArray.from_numpy((rng.standard_normal((3, 3, CI, CO)) * 0.1).astype(np.float32))
        b = DeviceArray.from_numpy(rng.standard_normal((CO,)).astype(np.float32))
        y = DeviceArray.zeros((B * H * W * CO,))
        dw = DeviceArray.zeros((3, 3, CI, CO))
        for what in ('fwd', 'wgrad'):
            for it in range(2):
                L.check(lib.dl4ds_profile_enable(1))
                for _ in range(reps):
                    if what == 'fwd':
                        L.check(lib.dl4ds_op_conv2d_fwd(x.ptr, w.ptr, b.ptr, None, y.ptr, B, H, W, CI, CO, 3, 0, 0))
                    else:
                        L.check(lib.dl4ds_op_conv2d_wgrad(x.ptr, y.ptr, dw.ptr, B, H, W, CI, CO, 3, 0, 0))
                buf = ctypes.create_string_buffer(1 << 16)
                L.check(lib.dl4ds_profile_report(buf, len(buf)))
                L.check(lib.dl4ds_profile_enable(0))
            rep = json.loads(buf.value.decode())
            main = max(rep.items(), key=lambda kv: kv[1]['ms'])
            ms = main[1]['ms'] / reps
            nbytes = 4.0 * B * H * W * (CI + CO)
            print(f'{CI}->{CO} {what:5s} N={B:4d} {H:4d}x{W:4d}  {ms:7.4f} ms  {nbytes / ms / 1e6:7.0f} GB/s  [{main[0]}]', flush=True)
        del x, y
