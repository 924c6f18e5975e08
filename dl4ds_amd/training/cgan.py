"""CGANTrainer -- same signature and ``.run()`` as dl4ds/training/cgan.py:30-444; ``train_step``,
``generator_loss`` and ``discriminator_loss`` (:525-639) execute inside libdl4ds_hip (csrc/cgan.hip)."""
import time

import numpy as np

from .. import POSTUPSAMPLING_METHODS
from ..dataloader import create_batch_hr_lr, DeviceDataGenerator
from .. import models as M
from .. import parallel
from .base import Trainer
from .engine import CGANEngine


class CGANTrainer(Trainer):
    def __init__(self, backbone, upsampling, data_train, data_test, data_train_lr=None, data_test_lr=None,
                 predictors_train=None, predictors_test=None, scale=5, patch_size=None, time_window=None, loss='mae',
                 epochs=60, batch_size=16, learning_rates=(2e-4, 2e-4), device='GPU', gpu_memory_growth=True,
                 model_list=None, steps_per_epoch=None, interpolation='inter_area', static_vars=None,
                 checkpoints_frequency=0, save=False, save_path=None, save_logs=False, save_loss_history=True,
                 generator_params={}, discriminator_params={}, verbose=True):
        super().__init__(backbone=backbone, upsampling=upsampling, data_train=data_train, data_train_lr=data_train_lr,
                         time_window=time_window, loss=loss, batch_size=batch_size, patch_size=patch_size, scale=scale,
                         device=device, gpu_memory_growth=gpu_memory_growth, verbose=verbose, model_list=model_list,
                         save=save, save_path=save_path)
        self.data_test, self.data_test_lr = data_test, data_test_lr
        self.predictors_train, self.predictors_test = predictors_train, predictors_test
        if self.predictors_train is not None and not isinstance(self.predictors_train, list):
            raise TypeError('`predictors_train` must be a list of ndarrays')
        if self.predictors_test is not None and not isinstance(self.predictors_test, list):
            raise TypeError('`predictors_test` must be a list of ndarrays')
        self.epochs, self.steps_per_epoch, self.interpolation = epochs, steps_per_epoch, interpolation
        self.static_vars = None if static_vars is None else [getattr(v, 'values', v) for v in static_vars]
        if self.static_vars is None:
            # the reference crashes here (cgan.py:354 `static_array=aux_hr` is unbound without static variables)
            raise ValueError('CGANTrainer needs at least one static variable (reference behaviour: cgan.py:343-354)')
        self.checkpoints_frequency, self.save_logs, self.save_loss_history = checkpoints_frequency, save_logs, save_loss_history
        self.generator_params, self.discriminator_params = generator_params, discriminator_params
        self.learning_rates = learning_rates
        if self.time_window is not None and not self.model_is_spatiotemporal:
            self.time_window = None
        self.gentotal, self.gengan, self.genpxloss, self.disc = [], [], [], []

    def setup_model(self):
        """cgan.py:174-262."""
        n_channels, n_aux = self._channels(self.predictors_train, self.static_vars)
        lr_size, hr_size = self._grid_sizes()
        gp = self.generator_params
        if self.upsampling in POSTUPSAMPLING_METHODS:
            if self.model_is_spatiotemporal:
                self.generator = M.recnet_postupsampling(backbone_block=self.backbone, upsampling=self.upsampling,
                                                         scale=self.scale, n_channels=n_channels, n_aux_channels=n_aux,
                                                         lr_size=lr_size, time_window=self.time_window, **gp)
            else:
                self.generator = M.net_postupsampling(backbone_block=self.backbone, upsampling=self.upsampling,
                                                      scale=self.scale, n_channels=n_channels, n_aux_channels=n_aux,
                                                      lr_size=lr_size, **gp)
        elif self.model_is_spatiotemporal:
            self.generator = M.recnet_pin(backbone_block=self.backbone, n_channels=n_channels, n_aux_channels=n_aux,
                                          hr_size=hr_size, time_window=self.time_window, **gp)
        elif self.backbone == 'unet':
            self.generator = M.unet_pin(backbone_block=self.backbone, n_channels=n_channels, n_aux_channels=n_aux,
                                        hr_size=hr_size, **gp)
        else:
            self.generator = M.net_pin(backbone_block=self.backbone, n_channels=n_channels, n_aux_channels=n_aux,
                                       hr_size=hr_size, **gp)
        self.discriminator = M.residual_discriminator(n_channels=n_channels, upsampling=self.upsampling,
                                                      is_spatiotemporal=self.model_is_spatiotemporal, scale=self.scale,
                                                      lr_size=lr_size, hr_size=hr_size, time_window=self.time_window,
                                                      **self.discriminator_params)
        if self.verbose == 1 and self.running_on_first_worker:
            self.generator.summary(line_length=150)
            self.discriminator.summary(line_length=150)

    def run(self):
        """cgan.py:264-444: epoch/step loop around train_step."""
        t0 = time.time()
        self.setup_model()
        # genlr, dislr = learning_rates; a float / 1-tuple sets both (cgan.py:271-278)
        self.engine = CGANEngine(self.generator, self.discriminator, loss=self.lossf,
                                 learning_rate=self.learning_rates, beta_1=0.5)
        n_samples = self.data_train.shape[0] - (self.time_window or 0)
        if self.steps_per_epoch is None:
            self.steps_per_epoch = n_samples // self.batch_size
        rng = np.random.default_rng(17)
        preds = None if self.predictors_train is None else np.concatenate(self.predictors_train, axis=-1)
        # dataset in HBM, batches gathered by csrc/batchprep.hip (same crops as the numpy loop: it draws them from the
        # same generator), a caller-supplied LR array included; the host loop below remains for models without static variables
        # (whose batches carry no auxiliary array)
        dev = None
        if self.static_vars is not None:
            dev = DeviceDataGenerator(self.data_train, self.data_train_lr, self.backbone, self.upsampling, self.scale,
                                      batch_size=self.batch_size, patch_size=self.patch_size,
                                      time_window=self.time_window, static_vars=self.static_vars,
                                      predictors=self.predictors_train, interpolation=self.interpolation)
            dev.rng = rng
        first = True
        for epoch in range(self.epochs):
            idx = parallel.shard_indices(n_samples, self.rank, self.world, seed=17, epoch=epoch)
            steps = min(self.steps_per_epoch // self.world, len(idx) // self.batch_size)
            for i in range(steps):
                if dev is not None:
                    (lr_dev, aux_dev), (hr_dev,) = dev.prepare(idx[i * self.batch_size:(i + 1) * self.batch_size])
                    losses = self.engine.step_device([lr_dev.ptr, aux_dev.ptr], hr_dev.ptr, self.batch_size,
                                                     want_losses=True)
                    if first and self.world > 1:
                        parallel.broadcast_trainer(self.engine)
                    first = False
                    for lst, v in zip((self.gentotal, self.gengan, self.genpxloss, self.disc), losses):
                        lst.append(v)
                    continue
                (lr_array, aux_hr), (hr_array,) = create_batch_hr_lr(
                    idx, i, self.data_train, self.data_train_lr, upsampling=self.upsampling, scale=self.scale,
                    batch_size=self.batch_size, patch_size=self.patch_size, time_window=self.time_window,
                    static_vars=self.static_vars, predictors=preds, interpolation=self.interpolation, rng=rng)
                losses = self.engine.step([lr_array, aux_hr], hr_array)
                if first and self.world > 1:
                    parallel.broadcast_trainer(self.engine)       # cgan.py:626-637 (after the first step)
                first = False
                for lst, v in zip((self.gentotal, self.gengan, self.genpxloss, self.disc), losses):
                    lst.append(v)
            # cgan.py:370-377: a checkpoint (both models + both optimisers) and the generator's weights every
            # `checkpoints_frequency` epochs, written by the first worker only
            if self.checkpoints_frequency > 0 and self.running_on_first_worker and (epoch + 1) % self.checkpoints_frequency == 0:
                self._save_checkpoint(epoch + 1)
            if self.verbose and self.running_on_first_worker and steps:
                print(f'Epoch {epoch + 1}/{self.epochs} - gen_total {self.gentotal[-1]:.4f} gen_gan {self.gengan[-1]:.4f} '
                      f'gen_px {self.genpxloss[-1]:.4f} disc {self.disc[-1]:.4f}')
        if self.checkpoints_frequency > 0 and self.running_on_first_worker:          # cgan.py:379-382: the last state
            self._save_checkpoint(self.epochs)
        self._test_loss()
        self.running_time = time.time() - t0
        if self.save_loss_history and self.save and self.running_on_first_worker:
            np.save(self.save_path + 'losses.npy', np.array([self.gentotal, self.gengan, self.genpxloss, self.disc]))
        self.save_results(self.generator, folder_prefix='cgan_')
        return self

    def _test_loss(self, max_batch=None):
        """cgan.py:386-440: pixel loss of generator.predict on the whole test set, on the first worker only (no
        collective is involved, so the other ranks simply skip it).  The reference builds ONE batch of n_test samples;
        here the same samples go through the generator in chunks of `batch_size` and the per-chunk losses are averaged
        with their sample counts (identical for the mean-type pixel losses; for the DSSIM family, whose dynamic range is
        taken over the batch, it is the mean of per-chunk values)."""
        if self.data_test is None or not self.running_on_first_worker:
            return
        data_test = np.asarray(getattr(self.data_test, 'values', self.data_test))
        data_test_lr = None if self.data_test_lr is None else np.asarray(getattr(self.data_test_lr, 'values', self.data_test_lr))
        preds = None if self.predictors_test is None else np.concatenate(self.predictors_test, axis=-1)
        n_test = data_test.shape[0] - (self.time_window or 0)
        if n_test <= 0:
            return
        rng = np.random.default_rng(23)
        idx = rng.permutation(n_test)
        from .engine import SupervisedEngine
        ev = SupervisedEngine(self.generator, loss=self.lossf, learning_rate=1e-3)     # evaluate() only: no update is made
        bs = int(max_batch or self.batch_size)
        tot, cnt = 0.0, 0
        for i in range(0, n_test, bs):
            chunk = idx[i:i + bs]
            (lr_array, aux_hr), (hr_array,) = create_batch_hr_lr(
                chunk, 0, data_test, data_test_lr, upsampling=self.upsampling, scale=self.scale, batch_size=len(chunk),
                patch_size=self.patch_size, time_window=self.time_window, static_vars=self.static_vars, predictors=preds,
                interpolation=self.interpolation, rng=rng)
            tot += ev.evaluate([lr_array, aux_hr], hr_array) * len(chunk)
            cnt += len(chunk)
        self.test_loss = tot / cnt
        if self.verbose:
            print(f'\n{self.lossf} on the test set: {self.test_loss}')

    def _save_checkpoint(self, epoch):
        import os
        d = os.path.join(getattr(self, 'savecheckpoint_path', None) or self.save_path or './', 'checkpoints')
        os.makedirs(d, exist_ok=True)
        self.engine.save_checkpoint(os.path.join(d, f'checkpoint_epoch-{epoch}.npz'))
        np.savez(os.path.join(d, f'save_epoch{epoch}_generator_weights.npz'), **self.generator.get_weights())

    fit = run


# ------------------------------------------------------------------------------------------------------------------
# The reference's module-level functions of this file (cgan.py:447-639).  The arithmetic of all of them runs inside
# libdl4ds_hip (csrc/cgan.hip, csrc/losses.hip); they are thin, same-named entry points over CGANEngine so that code
# written against ``dl4ds.training.cgan`` finds them here.
def _loss_name(f):
    name = f if isinstance(f, str) else getattr(f, '__name__', None)
    from ..ops import LOSS_KINDS
    if name not in LOSS_KINDS:
        raise ValueError(f'`gen_pxloss_function` must be one of dl4ds_amd.losses ({sorted(LOSS_KINDS)}), got {f!r}')
    return name


def generator_loss(disc_generated_output, gen_output, target, gen_pxloss_function, lambda_scaling_factor=100):
    """cgan.py:525-553: (total, gan, px) with gan = BCE(ones, D(x, G(x))) and total = gan + lambda * px."""
    from .. import ops
    gan_loss, _ = ops.bce(np.asarray(disc_generated_output, np.float32), 1.0)
    px_loss, _ = ops.loss(_loss_name(gen_pxloss_function), np.asarray(target, np.float32), np.asarray(gen_output, np.float32),
                          want_grad=False)
    return gan_loss + lambda_scaling_factor * px_loss, gan_loss, px_loss


def discriminator_loss(disc_real_output, disc_generated_output):
    """cgan.py:556-572: BCE(ones, D(x, y)) + BCE(zeros, D(x, G(x)))."""
    from .. import ops
    real_loss, _ = ops.bce(np.asarray(disc_real_output, np.float32), 1.0)
    generated_loss, _ = ops.bce(np.asarray(disc_generated_output, np.float32), 0.0)
    return real_loss + generated_loss


class _OptimizerHandle:
    """Stands in for the tf.keras.optimizers.Adam objects the reference threads through train_step / load_checkpoint:
    both Adam states live inside one CGANEngine; the handle says which of them it is."""

    def __init__(self, engine, which):
        self.engine, self.which = engine, which

    @property
    def iterations(self):
        return self.engine.optimizer_state(self.which)[2]

    def variables(self):
        m, v, step = self.engine.optimizer_state(self.which)
        return [np.int64(step)] + [a for pair in zip(m.values(), v.values()) for a in pair]


def make_optimizers(generator, discriminator, gen_pxloss_function='mae', learning_rates=(2e-4, 2e-4), beta_1=0.5):
    """The two Adam(lr, beta_1=0.5) optimisers of cgan.py:271-278 as handles on one CGANEngine."""
    eng = CGANEngine(generator, discriminator, loss=_loss_name(gen_pxloss_function), learning_rate=learning_rates, beta_1=beta_1)
    return _OptimizerHandle(eng, 'generator'), _OptimizerHandle(eng, 'discriminator')


def train_step(lr_array, hr_array, generator, discriminator, generator_optimizer, discriminator_optimizer, epoch=0,
               gen_pxloss_function='mae', summary_writer=None, first_batch=False, static_array=None):
    """cgan.py:575-639 with the same argument list: one simultaneous generator + discriminator update; returns
    (gen_total_loss, gen_gan_loss, gen_px_loss, disc_loss).  The optimisers are the handles of ``make_optimizers`` (or of
    ``load_checkpoint``); with several ranks the first batch broadcasts variables and optimiser slots from rank 0."""
    eng = generator_optimizer.engine
    if eng is not discriminator_optimizer.engine or eng.generator is not generator or eng.discriminator is not discriminator:
        raise ValueError('train_step: the optimisers must come from make_optimizers / load_checkpoint for these two models')
    inputs = [lr_array] if static_array is None else [lr_array, static_array]
    out = eng.step(inputs, hr_array)
    if first_batch and parallel.rank_world_from_env()[1] > 1 and parallel.is_initialized():
        parallel.broadcast_trainer(eng)              # cgan.py:626-637
    return out


def load_checkpoint(checkpoint_dir, checkpoint_number, backbone, upsampling, scale, input_height_width, n_static_vars=0,
                    n_predictors=0, time_window=None, n_blocks=(20, 4), n_filters=(8, 32), attention=False,
                    localcon_layer=False):
    """cgan.py:447-522: rebuild generator + discriminator with the same arguments, restore the checkpoint
    ``<checkpoint_dir>/checkpoint_epoch-<checkpoint_number>`` (the .npz this package's CGANTrainer writes,
    training/cgan.py::_save_checkpoint) into them and both Adam states; returns
    (generator, generator_optimizer, discriminator, discriminator_optimizer)."""
    import os
    n_channels, n_aux_channels = 1, 0
    if n_static_vars > 0:
        n_channels += n_static_vars
        n_aux_channels += n_static_vars
    if n_predictors > 0:
        n_channels += n_predictors
    spt = time_window is not None and time_window > 1
    gkw = dict(n_filters=n_filters[0], n_blocks=n_blocks[0], n_channels_out=1, attention=attention, localcon_layer=localcon_layer)
    if upsampling in POSTUPSAMPLING_METHODS:
        if spt:
            generator = M.recnet_postupsampling(backbone_block=backbone, upsampling=upsampling, scale=scale, n_channels=n_channels,
                                                n_aux_channels=n_aux_channels, lr_size=input_height_width, time_window=time_window,
                                                **gkw)
        else:
            generator = M.net_postupsampling(backbone_block=backbone, upsampling=upsampling, scale=scale, n_channels=n_channels,
                                             n_aux_channels=n_aux_channels, lr_size=input_height_width, **gkw)
        lr_size = tuple(input_height_width)
        hr_size = (int(lr_size[0] * scale), int(lr_size[1] * scale))
    elif upsampling == 'pin':
        if spt:
            generator = M.recnet_pin(backbone_block=backbone, n_channels=n_channels, n_aux_channels=n_aux_channels,
                                     hr_size=input_height_width, time_window=time_window, **gkw)
        else:
            generator = M.net_pin(backbone_block=backbone, n_channels=n_channels, n_aux_channels=n_aux_channels,
                                  hr_size=input_height_width, **gkw)
        hr_size = tuple(input_height_width)
        lr_size = hr_size
    else:
        raise ValueError(f'unknown upsampling {upsampling}')
    discriminator = M.residual_discriminator(n_channels=n_channels, upsampling=upsampling, is_spatiotemporal=spt, scale=scale,
                                             lr_size=lr_size, hr_size=hr_size, time_window=time_window, n_filters=n_filters[1],
                                             n_res_blocks=n_blocks[1], attention=attention)
    gopt, dopt = make_optimizers(generator, discriminator, 'mae', (2e-4, 2e-4), beta_1=0.5)       # cgan.py:513-514
    gopt.engine.load_checkpoint(os.path.join(checkpoint_dir, f'checkpoint_epoch-{checkpoint_number}.npz'))
    return generator, gopt, discriminator, dopt
