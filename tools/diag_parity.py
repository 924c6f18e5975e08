"""Diagnostic (GPU): per-tensor gradient deviation of the HIP path and of the torch-fp32 oracle from the fp64 oracle.
    python tools/diag_parity.py cfg1|cfg5g|cfg4 [B] [seed]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.set_num_threads(32)
import dl4ds_amd.models as PM
from dl4ds_amd.training import SupervisedEngine
from oracle import torch_ops as T, models as M, train as TR

which = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1001
loss = sys.argv[4] if len(sys.argv) > 4 else 'mae'
rng = np.random.default_rng(seed)
if which == 'cfg1':
    model = PM.net_pin('resnet', 2, 0, hr_size=(128, 128), seed=11)
    kind, ocfg = 'net_pin', dict(backbone_block='resnet')
    xs, ss, ys = (B, 128, 128, 2), None, (B, 128, 128, 1)
elif which == 'cfg5g':
    model = PM.unet_pin('unet', 5, 1, hr_size=(512, 512), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=3)
    kind, ocfg = 'unet_pin', dict(n_filters=8, n_blocks=6, decoder_upsampling='dc')
    xs, ss, ys = (B, 512, 512, 5), (B, 512, 512, 1), (B, 512, 512, 1)
elif which == 'cfg4':
    model = PM.recnet_postupsampling('densenet', 'rc', 4, 1, 1, (64, 64), time_window=8, attention=True, localcon_layer=True, seed=3)
    kind, ocfg = 'recnet_postupsampling', dict(backbone_block='densenet', upsampling='rc', scale=4, time_window=8, attention=True, localcon_layer=True)
    xs, ss, ys = (B, 8, 64, 64, 1), (B, 256, 256, 1), (B, 8, 256, 256, 1)
w = model.get_weights()
r2 = np.random.default_rng(9)
for k in w:
    if k.endswith('bias'):
        w[k] = (r2.standard_normal(w[k].shape) * 0.05).astype(np.float32)
model.set_weights(w)
x = rng.standard_normal(xs).astype(np.float32)
s = None if ss is None else rng.standard_normal(ss).astype(np.float32)
y = rng.standard_normal(ys).astype(np.float32)
ins = [x] if s is None else [x, s]
eng = SupervisedEngine(model, loss=loss, learning_rate=1e-3)
l_hip, g_hip = eng.loss_and_grads(ins, y)
out = model(ins)
res = {}
for dt in (np.float64, np.float32):
    P = M.Params()
    for k, v in w.items():
        P[k] = v.astype(dt)
    PT = M.convert(P, T, requires_grad=True)
    lv, g, pred = TR.supervised_step(kind, ocfg, PT, T.asarray(x.astype(dt)), None if s is None else T.asarray(s.astype(dt)),
                                     T.asarray(y.astype(dt)), loss=loss)
    res[dt] = (lv, {k: v.numpy().astype(np.float64) for k, v in g.items()}, pred.numpy().astype(np.float64))
lv, g64, p64 = res[np.float64]
_, g32, p32 = res[np.float32]
print('loss hip %.9f f64 %.9f f32 %.9f' % (l_hip, lv, res[np.float32][0]))
e = np.abs(out - p64); e32 = np.abs(p32 - p64)
print('fwd: hip max err %.3e (at %s)  f32 max err %.3e   scale %.3e' % (e.max(), np.unravel_index(e.argmax(), e.shape), e32.max(), np.abs(p64).max()))
print('sign flips of (pred - y): hip %d  f32 %d' % ((np.sign(out - y) != np.sign(p64 - y)).sum(), (np.sign(p32 - y) != np.sign(p64 - y)).sum()))
gs = max(np.abs(v).max() for v in g64.values())
print('%-44s %10s %10s %10s %10s' % ('tensor', 'max|g|', 'hip rel', 'f32 rel', 'hip/gscale'))
for k in g64:
    r = np.abs(g64[k]).max()
    dh = np.abs(g_hip[k] - g64[k]).max()
    d3 = np.abs(g32[k] - g64[k]).max()
    flag = ' <<<' if dh > 1e-3 * r else ''
    print('%-44s %10.3e %10.2e %10.2e %10.2e%s' % (k[:44], r, dh / max(r, 1e-300), d3 / max(r, 1e-300), dh / gs, flag))
