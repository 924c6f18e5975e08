"""End-to-end SupervisedTrainer at the headline size (resnet + spc x4, 512^2 HR fields, batch 64): wall-clock per training
epoch including the on-device batch preparation and the Python loop, against the bare train-step rate of bench.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from dl4ds_amd.training import SupervisedTrainer
import dl4ds_amd.training.supervised as S

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 512
rng = np.random.default_rng(0)
mk = lambda n: rng.random((n, 512, 512, 1), dtype=np.float32)
tr, va, te = mk(n_train), mk(64), mk(64)
t = SupervisedTrainer('resnet', 'spc', tr, va, te, scale=4, batch_size=64, epochs=4, verbose=False, save=False)
orig = t._epoch_loss
marks = []
def timed(ds, steps, train):
    t0 = time.perf_counter()
    r = orig(ds, steps, train)
    marks.append((train, r[1], time.perf_counter() - t0))
    return r
t._epoch_loss = timed
t.run()
for train, n, dt in marks:
    print(f"{'train' if train else 'eval '} {n:3d} batches  {1e3 * dt / max(n, 1):8.2f} ms/batch  {64 * n / dt:8.0f} samples/s")
