"""cfg4 on the bench workload: the ConvLSTM bias gradients entry by entry, HIP path vs fp64 oracle (gate order i, f, c, o x 8 filters)."""
import os, sys
os.environ['DL4DS_TEST_HOOKS'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
import dl4ds_amd.models as PM
from dl4ds_amd.training import SupervisedEngine
from tests.parity import oracle_reference
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = PM.recnet_postupsampling('densenet', 'rc', 4, 1, 1, (64, 64), time_window=8, attention=True, localcon_layer=True, seed=7)
w = model.get_weights()
x, aux, y = bench.synthetic_batch_cfg4(1004, B)
eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
l_hip, g = eng.loss_and_grads([x, aux], y)
cfg = dict(backbone_block='densenet', upsampling='rc', scale=4, time_window=8, attention=True, localcon_layer=True)
ref = oracle_reference('supervised', 'recnet_postupsampling', cfg, w, x, aux, y, loss='mae', workers=min(B, 8))
np.set_printoptions(linewidth=200, precision=3)
gs = max(np.abs(v).max() for v in ref['grads'].values())
print('loss', l_hip, ref['loss'], 'gscale', gs)
for k in ref['grads']:
    if 'convlstm' in k and not k.endswith('bias') and ('Block5' in k or 'Block3' in k or 'Block1' in k):
        r, h = np.asarray(ref['grads'][k]), np.asarray(g[k])
        F = r.shape[-1] // 4
        print(k, r.shape, 'own', np.abs(r).max(), ' per gate (max |ref|, max |hip - ref|):',
              [(float(np.abs(r[..., F * i:F * i + F]).max()), float(np.abs(h - r)[..., F * i:F * i + F].max())) for i in range(4)])
    if 'convlstm' in k and k.endswith('bias') and 'Block5' in k:
        r, h = np.asarray(ref['grads'][k]), np.asarray(g[k])
        print(k, 'own', np.abs(r).max())
        for gi, gn in enumerate('ifco'):
            sl = slice(8 * gi, 8 * gi + 8)
            print('  ', gn, 'ref', r[sl], '\n      hip', h[sl], '\n      band', np.asarray(ref['band'][k])[sl], 'noise', np.asarray(ref['noise'][k])[sl])
        print('   weights bias', np.asarray(w[k]))
