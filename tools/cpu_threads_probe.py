import sys, time, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from oracle import models as M, train as TR, torch_ops as T
from bench import synthetic_batch
cfg = dict(backbone_block='resnet', upsampling='spc', scale=4)
P0 = M.init_params('net_postupsampling', (1, 8, 8, 1), **cfg)
for b, threads in ((2, 256), (2, 64), (2, 32), (2, 16), (4, 64), (8, 128)):
    torch.set_num_threads(threads)
    P = M.convert(P0, T, requires_grad=True)
    x, y = synthetic_batch(1, b)
    xt, yt = torch.from_numpy(x), torch.from_numpy(y)
    TR.supervised_step('net_postupsampling', cfg, P, xt, None, yt)
    t0 = time.perf_counter(); TR.supervised_step('net_postupsampling', cfg, P, xt, None, yt); dt = time.perf_counter() - t0
    print(b, threads, f'{dt:.2f}s', f'{b/dt:.3f} samples/s', flush=True)
