"""Winograd vs direct 3x3 kernels on the headline model's MFMA-bound layers (B = 64), each timed in isolation with the
library profiler: python tools/wino_microbench.py [reps]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
lib = L.lib()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(os.environ.get('MB_N', 64))
rng = np.random.default_rng(0)
CASES = [  # H, Cin, Cout, d2s
    (128, 48, 48, 0), (128, 48, 192, 2), (256, 48, 32, 2), (128, 40, 48, 0), (128, 32, 32, 0), (128, 24, 32, 0)]
if os.environ.get('MB_CASES'):          # e.g. MB_CASES=1,0: only those rows of CASES
    CASES = [CASES[int(i)] for i in os.environ['MB_CASES'].split(',')]
MODES = tuple(os.environ.get('MB_MODES', 'wino,direct').split(','))
for H, CI, CO, r in CASES:
    x = DeviceArray.from_numpy(rng.standard_normal((B, H, H, CI)).astype(np.float32))
    w = DeviceArray.from_numpy((rng.standard_normal((3, 3, CI, CO)) * 0.1).astype(np.float32))
    b = DeviceArray.from_numpy(rng.standard_normal((CO,)).astype(np.float32))
    rr = max(r, 1)
    y = DeviceArray.zeros((B, rr * H, rr * H, CO // (rr * rr)))
    dz = DeviceArray.from_numpy(rng.standard_normal((B, rr * H, rr * H, CO // (rr * rr))).astype(np.float32))
    dx = DeviceArray.zeros((B, H, H, CI))
    for mode in MODES:
        if mode == 'direct':
            os.environ['DL4DS_NO_WINOGRAD'] = '1'
        else:
            os.environ.pop('DL4DS_NO_WINOGRAD', None)
        if mode.startswith('f44'):          # F(4x4, 3x3) where it is built (run with DL4DS_TEST_HOOKS=1); 'f44force': also ragged cout chunks
            os.environ['DL4DS_WINO_F44'] = 'force' if mode == 'f44force' else '1'
        else:
            os.environ.pop('DL4DS_WINO_F44', None)
        for what in ('fwd', 'dgrad'):
            L.check(lib.dl4ds_profile_enable(1))
            for _ in range(reps):
                if what == 'fwd':
                    L.check(lib.dl4ds_op_conv2d_fwd(x.ptr, w.ptr, b.ptr, None, y.ptr, B, H, H, CI, CO, 3, 1, r))
                else:
                    L.check(lib.dl4ds_op_conv2d_dgrad(dz.ptr, w.ptr, dx.ptr, B, H, H, CI, CO, 3, r, 0))
            buf = ctypes.create_string_buffer(1 << 16)
            L.check(lib.dl4ds_profile_report(buf, len(buf)))
            L.check(lib.dl4ds_profile_enable(0))
            rep = json.loads(buf.value.decode())
            tot = sum(v['ms'] for v in rep.values()) / reps
            direct_tf = 2.0 * B * H * H * 9 * CI * CO / (tot * 1e-3) / 1e12
            tags = ' + '.join(f"{k} x{v['n'] // reps}" for k, v in rep.items())
            print(f'{H:4d}^2 {CI:3d}->{CO:3d} d2s={r} {mode:6s} {what:5s} {tot:8.4f} ms  {direct_tf:6.1f} direct-equivalent TFLOP/s  [{tags}]', flush=True)

# ---- weight gradients
if os.environ.get('MB_NO_WGRAD'):
    sys.exit(0)
print('weight gradients', flush=True)
for H, CI, CO, r in CASES:
    x = DeviceArray.from_numpy(rng.standard_normal((B, H, H, CI)).astype(np.float32))
    rr = max(r, 1)
    dz = DeviceArray.from_numpy(rng.standard_normal((B, rr * H, rr * H, CO // (rr * rr))).astype(np.float32))
    dw = DeviceArray.zeros((3, 3, CI, CO))
    for mode in MODES:
        if mode == 'direct':
            os.environ['DL4DS_NO_WINOGRAD'] = '1'
        else:
            os.environ.pop('DL4DS_NO_WINOGRAD', None)
        for it in range(2):          # (the first round warms up: attributes, scratch)
            L.check(lib.dl4ds_profile_enable(1))
            for _ in range(reps):
                L.check(lib.dl4ds_op_conv2d_wgrad(x.ptr, dz.ptr, dw.ptr, B, H, H, CI, CO, 3, r, 0))
            buf = ctypes.create_string_buffer(1 << 16)
            L.check(lib.dl4ds_profile_report(buf, len(buf)))
            L.check(lib.dl4ds_profile_enable(0))
        rep = json.loads(buf.value.decode())
        tot = sum(v['ms'] for v in rep.values()) / reps
        direct_tf = 2.0 * B * H * H * 9 * CI * CO / (tot * 1e-3) / 1e12
        tags = ' + '.join(f"{k} x{v['n'] // reps} {v['ms'] / reps:.3f}" for k, v in rep.items())
        print(f'{H:4d}^2 {CI:3d}->{CO:3d} d2s={r} {mode:6s} wgrad {tot:8.4f} ms  {direct_tf:6.1f} direct-equivalent TFLOP/s  [{tags}]', flush=True)
