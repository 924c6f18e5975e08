"""Where does the F(4x4) kernel differ from the oracle?  Error maps by pixel (mod 4), tile and channel for structured inputs."""
import os, sys
os.environ['DL4DS_TEST_HOOKS'] = '1'
os.environ['DL4DS_WINO_FORCE'] = 'all'
os.environ['DL4DS_WINO_F44'] = 'force'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dl4ds_amd import ops
from oracle import np_ops as N
rng = np.random.default_rng(0)


def report(name, got, ref):
    err = np.abs(got - ref)
    sc = np.abs(ref).max()
    print(f'--- {name}: max err {err.max() / sc:.3e}')
    if err.max() / sc < 1e-4:
        return
    n, h, w, c = err.shape
    e = err[0]
    print('by (row % 4, col % 4):')
    for i in range(4):
        print('   ', ' '.join(f'{e[i::4, j::4].max() / sc:9.2e}' for j in range(4)))
    print('by tile (16 px groups, first 16x16):')
    for ty in range(min(4, h // 4)):
        print('   ', ' '.join(f'{e[4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4].max() / sc:9.2e}' for tx in range(min(4, w // 4))))
    print('by channel:', ' '.join(f'{e[..., k].max() / sc:8.1e}' for k in range(c)))



h = w = 16; ci, co = 48, 32
wt = (rng.standard_normal((3, 3, ci, co)) * 0.2).astype(np.float32)
for lo in (0, 16, 32):
    x = np.zeros((1, h, w, ci), np.float32)
    x[..., lo:lo + 16] = rng.standard_normal((1, h, w, 16))
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), None)
    got = ops.conv2d(x, wt, None)
    report(f'only channels {lo}..{lo + 15}', got, ref)
for lo in (0, 4, 8, 12):
    x = np.zeros((1, h, w, ci), np.float32)
    x[..., lo:lo + 4] = rng.standard_normal((1, h, w, 4))
    ref = N.conv2d(x.astype(np.float64), wt.astype(np.float64), None)
    got = ops.conv2d(x, wt, None)
    report(f'only channels {lo}..{lo + 3}', got, ref)
