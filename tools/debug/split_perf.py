import os, sys
os.environ['DL4DS_TEST_HOOKS'] = '1'
sys.path.insert(0, '.')
import numpy as np
import dl4ds_amd.ops as ops
rng = np.random.default_rng(0)
n, h, w, ci, co = 64, 128, 128, 48, 48
x = rng.standard_normal((n, h, w, ci)).astype(np.float32)
wt = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
b = rng.standard_normal(co).astype(np.float32)
os.environ['DL4DS_SPLIT_TRACE'] = '1'
got = ops.conv2d(x, wt, b)
