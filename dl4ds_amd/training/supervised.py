"""SupervisedTrainer -- same signature and ``.run()`` as dl4ds/training/supervised.py:28-416.  Keras'
compile/fit/evaluate become an explicit epoch/step loop around libdl4ds_hip's fused train step; Horovod's
DistributedOptimizer / BroadcastGlobalVariablesCallback become one RCCL all-reduce per step and one
broadcast before the first step."""
import os
import time

import numpy as np

from .. import POSTUPSAMPLING_METHODS
from ..dataloader import DataGenerator, DeviceDataGenerator
from .. import models as M
from .. import parallel
from .base import Trainer
from .engine import SupervisedEngine


class SupervisedTrainer(Trainer):
    def __init__(self, backbone, upsampling, data_train, data_val, data_test, data_train_lr=None, data_val_lr=None,
                 data_test_lr=None, predictors_train=None, predictors_val=None, predictors_test=None,
                 static_vars=None, scale=5, interpolation='inter_area', patch_size=None, time_window=None,
                 batch_size=64, loss='mae', epochs=60, steps_per_epoch=None, test_steps=None, validation_steps=None,
                 device='GPU', gpu_memory_growth=True, use_multiprocessing=False, model_list=None,
                 learning_rate=(1e-3, 1e-4), lr_decay_after=1e5, early_stopping=False, patience=6, min_delta=0,
                 show_plot=True, save=False, save_path=None, save_bestmodel=False, trained_model=None,
                 trained_epochs=0, verbose=True, checkpoint=None, device_data=True, **architecture_params):
        super().__init__(backbone=backbone, upsampling=upsampling, data_train=data_train, data_train_lr=data_train_lr,
                         time_window=time_window, loss=loss, batch_size=batch_size, patch_size=patch_size, scale=scale,
                         device=device, gpu_memory_growth=gpu_memory_growth, use_multiprocessing=use_multiprocessing,
                         verbose=verbose, model_list=model_list, save=save, save_path=save_path, show_plot=show_plot)
        self.data_val, self.data_test = data_val, data_test
        self.data_val_lr, self.data_test_lr = data_val_lr, data_test_lr
        for name, p in (('predictors_train', predictors_train), ('predictors_test', predictors_test),
                        ('predictors_val', predictors_val)):
            if p is not None and not isinstance(p, list):
                raise TypeError(f'`{name}` must be a list of ndarrays')
        self.predictors_train, self.predictors_val, self.predictors_test = predictors_train, predictors_val, predictors_test
        self.static_vars = None if static_vars is None else [getattr(v, 'values', v) for v in static_vars]
        self.interpolation, self.epochs, self.steps_per_epoch = interpolation, epochs, steps_per_epoch
        self.validation_steps, self.test_steps = validation_steps, test_steps
        self.learning_rate, self.lr_decay_after = learning_rate, lr_decay_after
        self.early_stopping, self.patience, self.min_delta = early_stopping, patience, min_delta
        self.architecture_params = architecture_params
        self.trained_model, self.trained_epochs, self.save_bestmodel = trained_model, trained_epochs, save_bestmodel
        self.checkpoint, self.device_data = checkpoint, device_data

    def setup_datagen(self):
        """supervised.py:220-240."""
        kw = dict(backbone=self.backbone, upsampling=self.upsampling, scale=self.scale,
                  batch_size=self.global_batch_size, static_vars=self.static_vars, patch_size=self.patch_size,
                  interpolation=self.interpolation, time_window=self.time_window, rank=self.rank, world=self.world)
        def make(data, data_lr, predictors, seed):
            # datasets live in HBM and batches are gathered on the device (csrc/batchprep.hip: block means for the default
            # 'inter_area', cv2 tap tables for the other interpolations, composed gather passes for a caller-supplied LR array /
            # predictors on the LR grid / field sizes `scale` does not divide); device_data=False keeps the numpy loop
            if getattr(self, 'device_data', True):
                return DeviceDataGenerator(data, data_lr, predictors=predictors, seed=seed, **kw)
            return DataGenerator(data, data_lr, predictors=predictors, seed=seed, **kw)
        self.ds_train = make(self.data_train, self.data_train_lr, self.predictors_train, 1)
        self.ds_val = make(self.data_val, self.data_val_lr, self.predictors_val, 2)
        self.ds_test = make(self.data_test, self.data_test_lr, self.predictors_test, 3)

    def setup_model(self):
        """supervised.py:242-325."""
        n_channels, n_aux = self._channels(self.predictors_train, self.static_vars)
        lr_size, hr_size = self._grid_sizes()
        if self.trained_model is not None:
            self.model = self.trained_model
            print('Loading pre-trained model')
            return
        ap = self.architecture_params
        if self.upsampling in POSTUPSAMPLING_METHODS:
            if self.model_is_spatiotemporal:
                self.model = M.recnet_postupsampling(backbone_block=self.backbone, upsampling=self.upsampling,
                                                     scale=self.scale, n_channels=n_channels, n_aux_channels=n_aux,
                                                     lr_size=lr_size, time_window=self.time_window, **ap)
            else:
                self.model = M.net_postupsampling(backbone_block=self.backbone, upsampling=self.upsampling,
                                                  scale=self.scale, lr_size=lr_size, n_channels=n_channels,
                                                  n_aux_channels=n_aux, **ap)
        elif self.upsampling == 'pin':
            if self.model_is_spatiotemporal:
                self.model = M.recnet_pin(backbone_block=self.backbone, n_channels=n_channels, n_aux_channels=n_aux,
                                          hr_size=hr_size, time_window=self.time_window, **ap)
            elif self.backbone == 'unet':
                self.model = M.unet_pin(backbone_block=self.backbone, n_channels=n_channels, n_aux_channels=n_aux,
                                        hr_size=hr_size, **ap)
            else:
                self.model = M.net_pin(backbone_block=self.backbone, n_channels=n_channels, n_aux_channels=n_aux,
                                       hr_size=hr_size, **ap)
        if self.verbose == 1 and self.running_on_first_worker:
            self.model.summary(line_length=150)

    def _epoch_loss(self, ds, steps, train):
        """Mean loss over the epoch's batches.  Data parallel: every rank walks the same number of batches of its own
        shard (dataloader.equal_shard) and the epoch mean is averaged over the ranks (one RCCL all-reduce of two floats),
        so that early stopping and the reported val/test losses are the same everywhere -- ranks deciding on their own
        shard's val_loss would leave the epoch loop at different epochs and strand the others in the all-reduce."""
        n = len(ds) if steps is None else min(int(steps), len(ds))
        tot = 0.0
        for i in range(n):
            x, y = ds[i]
            if isinstance(ds, DeviceDataGenerator):
                ptrs, b = [a.ptr for a in x], ds.batch_size
                tot += (self.engine.step_device(ptrs, y[0].ptr, b, want_loss=True) if train
                        else self.engine.evaluate_device(ptrs, y[0].ptr, b))
            else:
                tot += self.engine.step(x, y[0]) if train else self.engine.evaluate(x, y[0])
        if self.world > 1:
            tot, cnt = parallel.allreduce_host([tot, float(n)], 'sum')
            return tot / max(cnt, 1.0), n
        return tot / max(n, 1), n

    def run(self):
        """supervised.py:328-416."""
        t0 = time.time()
        self.setup_datagen()
        self.setup_model()
        lr = self.learning_rate
        # Goyal et al. linear LR scaling by the number of workers (supervised.py:338-352)
        if isinstance(lr, (tuple, list)):
            lr = tuple(float(v) * self.world for v in lr)
        else:
            lr = float(lr) * self.world
        self.engine = SupervisedEngine(self.model, loss=self.lossf, learning_rate=lr, lr_decay_after=self.lr_decay_after)
        # resume: `checkpoint` (a file written by SupervisedEngine.save_checkpoint / a previous run with save=True)
        # restores weights, Adam slots and the iteration count, so that trained_epochs + remaining epochs continues the
        # same optimisation (the reference only re-loads the weights of `trained_model`, supervised.py:322-325)
        ckpt = getattr(self, 'checkpoint', None)
        if ckpt is not None:
            self.engine.load_checkpoint(ckpt)
        if self.world > 1:
            parallel.broadcast_trainer(self.engine)            # BroadcastGlobalVariablesCallback(0)
        steps = self.steps_per_epoch
        if steps is not None and self.world > 1:
            steps = steps // self.world                        # supervised.py:393-394
        hist = {'loss': [], 'val_loss': []}
        best, wait = np.inf, 0
        for epoch in range(self.trained_epochs, self.epochs):
            tl, n = self._epoch_loss(self.ds_train, steps, True)
            vl, _ = self._epoch_loss(self.ds_val, self.validation_steps, False)
            hist['loss'].append(tl)
            hist['val_loss'].append(vl)
            if self.verbose and self.running_on_first_worker:
                print(f'Epoch {epoch + 1}/{self.epochs} - {n} steps - loss: {tl:.6f} - val_loss: {vl:.6f}')
            # ModelCheckpoint(savecheckpoint_path/best_model, monitor='val_loss', save_best_only=False), first worker only
            # (supervised.py:379-390): with save_best_only=False Keras rewrites the file at the end of EVERY epoch; the
            # native format is the named-weights .npz + optimiser state of SupervisedEngine.save_checkpoint
            if self.save_bestmodel and self.running_on_first_worker:
                d = os.path.join(self.savecheckpoint_path, 'best_model')
                os.makedirs(d, exist_ok=True)
                self.engine.save_checkpoint(os.path.join(d, 'checkpoint.npz'))
                np.savez(os.path.join(d, 'model_weights.npz'), **self.model.get_weights())
                np.savetxt(os.path.join(d, 'epoch_val_loss.txt'), [epoch + 1, vl])
            if self.early_stopping:                            # EarlyStopping(monitor='val_loss', mode='min'); vl is rank-averaged
                if vl < best - self.min_delta:
                    best, wait = vl, 0
                else:
                    wait += 1
                    if wait >= self.patience:
                        if self.verbose and self.running_on_first_worker:
                            print(f'Epoch {epoch + 1}: early stopping')
                        break
        self.fithist = hist
        # model.evaluate(ds_test) -- the reference scores on the first worker only (supervised.py:409-411); here every rank
        # scores its shard of the test set and the mean over ranks is reported (all ranks must enter the collective)
        self.test_loss, _ = self._epoch_loss(self.ds_test, self.test_steps, False)
        if self.verbose and self.running_on_first_worker:
            print(f'\nScore on the test set: {self.test_loss}')
        self.running_time = time.time() - t0
        self.save_results(self.model)
        if self.save and self.running_on_first_worker and self.save_path is not None:
            self.engine.save_checkpoint(os.path.join(self.save_path, 'checkpoint.npz'))
        return self

    fit = run     # BASELINE.json calls it fit(); the reference method is run()
