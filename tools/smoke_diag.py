import sys, os
sys.path.insert(0,'/root/repo')
import numpy as np
import dl4ds_amd.models as PM
from oracle import torch_ops as T, models as M, train as TR
from oracle.reference import oracle_reference
cfg = dict(backbone_block='resnet', upsampling='spc', scale=4)
model = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (16, 16), seed=1)
rng = np.random.default_rng(0)
x = rng.standard_normal((2, 16, 16, 1)).astype(np.float32)
y = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
out = model([x])
P = M.Params()
for k, v in model.get_weights().items(): P[k] = v.astype(np.float64)
PT = M.convert(P, T)
p0 = TR.forward('net_postupsampling', cfg, PT, T.asarray(x.astype(np.float64)), None).numpy()
ref = oracle_reference('supervised', 'net_postupsampling', cfg, model.get_weights(), x, None, y, loss='mae')
sc = np.abs(p0).max()
print('hip vs plain fp64 oracle:', np.abs(out - p0).max() / sc)
print('banded mid vs plain fp64 oracle:', np.abs(ref['pred'] - p0).max() / sc)
print('hip vs banded mid:', np.abs(out - ref['pred']).max() / sc)
