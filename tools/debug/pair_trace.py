import os, sys, numpy as np, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import dl4ds_amd._lib as L
from dl4ds_amd.device import DeviceArray
lib = L.lib()
N, H, W = 32, 512, 512
rng = np.random.default_rng(0)
x = DeviceArray.from_numpy(rng.standard_normal((N, H, W, 8)).astype(np.float32))
w = DeviceArray.from_numpy((rng.standard_normal((3, 3, 8, 8)) * 0.1).astype(np.float32))
b = DeviceArray.from_numpy(rng.standard_normal((8,)).astype(np.float32))
y = DeviceArray.zeros((N, H, W, 8))
for _ in range(3):
    L.check(lib.dl4ds_op_conv2d_fwd(x.ptr, w.ptr, b.ptr, None, y.ptr, N, H, W, 8, 8, 3, 1, 0))
L.check(lib.dl4ds_sync())
