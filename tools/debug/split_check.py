import os, sys, numpy as np
sys.path.insert(0, '/root/repo')
from dl4ds_amd import ops
from oracle import np_ops as N
rng = np.random.default_rng(0)
for (n, h, w, ci, co) in [(2, 64, 64, 8, 8), (1, 37, 50, 5, 8), (2, 128, 128, 8, 4)]:
    x = rng.standard_normal((n, h, w, ci)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64))
    for relu in (False, True):
        got = ops.conv2d(x, wt, b, relu=relu)
        r = np.maximum(ref, 0) if relu else ref
        print(os.environ.get('DL4DS_PAIR_SPLIT', '0'), (n, h, w, ci, co), relu, 'max err / max', float(np.abs(got - r).max() / np.abs(r).max()))
