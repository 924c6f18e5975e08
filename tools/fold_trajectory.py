"""Loss trajectory of the headline model trained with the composed (folded) upsampling tail vs DL4DS_NO_FOLD=1 on the
same data, same initial weights, same Adam schedule: the two evaluate the same function, so the curves must track each
other up to the growth of fp32 rounding differences.   python tools/fold_trajectory.py [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import dl4ds_amd.models as PM
from dl4ds_amd.training import SupervisedEngine
from dl4ds_amd.device import DeviceArray
from bench import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
batches = [tuple(DeviceArray.from_numpy(a) for a in synthetic_batch(100 + i, B)) for i in range(4)]

def run(fold):
    if fold:
        os.environ.pop('DL4DS_NO_FOLD', None)
    else:
        os.environ['DL4DS_NO_FOLD'] = '1'
    m = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (128, 128), seed=7)
    e = SupervisedEngine(m, loss='mae', learning_rate=1e-3)
    out = []
    for i in range(steps):
        x, y = batches[i % 4]
        out.append(e.step_device([x.ptr], y.ptr, B, want_loss=True))
    return np.array(out), m.get_weights()

la, wa = run(True)
lb, wb = run(False)
# control: the unfolded graph again, with a different (equally valid) fp32 summation order in its largest kernels
os.environ['DL4DS_STREAM_NO_TALL'] = '1'
lc, wc = run(False)
os.environ.pop('DL4DS_STREAM_NO_TALL')
for i in list(range(0, steps, max(steps // 10, 1))) + [steps - 1]:
    print(f'step {i:4d}  folded {la[i]:.6f}  unfolded {lb[i]:.6f}  rel diff {abs(la[i] - lb[i]) / lb[i]:.2e}')
d = max(np.abs(wa[k] - wb[k]).max() for k in wa)
dc = max(np.abs(wc[k] - wb[k]).max() for k in wb)
print(f'control (unfolded, other tile shape vs unfolded): max |loss diff| / loss = {np.max(np.abs(lc - lb) / lb):.2e};  '
      f'max |weight diff| = {dc:.2e};  final loss {lc[-1]:.6f}')
print(f'loss {la[0]:.4f} -> {la[-1]:.4f};  max |loss diff| / loss = {np.max(np.abs(la - lb) / lb):.2e};  max |weight diff| = {d:.2e}')
