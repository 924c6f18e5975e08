"""Trainer base -- dl4ds/training/base.py:24-188: same constructor arguments, same validation, with the
Horovod pieces (hvd.init / local_rank pinning / rank-0 gating, :97-133) mapped to one process per MI355X and
an RCCL communicator (dl4ds_amd.parallel)."""
import os
from abc import ABC, abstractmethod

import numpy as np

from ..utils import check_compatibility_upsbackb, checkarg_loss
from .. import parallel


class Trainer(ABC):
    def __init__(self, backbone, upsampling, data_train, data_train_lr=None, time_window=None, loss='mae',
                 batch_size=64, patch_size=None, scale=4, device='GPU', gpu_memory_growth=True,
                 use_multiprocessing=False, verbose=True, model_list=None, save=True, save_path=None, show_plot=False):
        self.data_train = getattr(data_train, 'values', data_train)
        if not isinstance(self.data_train, np.ndarray):
            raise TypeError('`data_train` object must be of np.ndarray or xr.DataArray type')
        if not self.data_train.ndim > 3:
            raise ValueError('`data_train` must be at least 4D [samples, lat, lon, variables]')
        self.data_train_lr = None if data_train_lr is None else getattr(data_train_lr, 'values', data_train_lr)
        if self.data_train_lr is not None:
            if not isinstance(self.data_train_lr, np.ndarray):
                raise TypeError('`data_train_lr` must be a np.ndarray or xr.DataArray object')
            if self.data_train_lr.shape[0] != self.data_train.shape[0]:
                raise ValueError('`data_train_lr` and `data_train` must contain the same number of samples '
                                 '(equal 1st dim lenght)')
            if not self.data_train_lr.ndim > 3:
                raise ValueError('`data_train_lr` must be at least 4D [samples, lat, lon, variables]')
        self.backbone, self.upsampling = check_compatibility_upsbackb(backbone, upsampling, time_window)
        self.time_window = time_window
        self.model_is_spatiotemporal = bool(time_window is not None and time_window > 1)
        self.batch_size, self.patch_size, self.loss, self.scale = batch_size, patch_size, loss, scale
        self.device, self.gpu_memory_growth, self.use_multiprocessing = device, gpu_memory_growth, use_multiprocessing
        self.verbose, self.model_list, self.save = verbose, model_list, save
        self.save_path = './' if save_path is None else (save_path if save_path.endswith('/') else save_path + '/')
        self.savecheckpoint_path = self.save_path
        self.show_plot = show_plot
        if device != 'GPU':
            raise ValueError("device not recognized: dl4ds_amd runs on MI355X only (device='GPU'); there is no CPU path")
        # one process per GPU.  hvd.init() + set_visible_gpus(hvd.local_rank()) (base.py:97-107): bind LOCAL_RANK's GPU and
        # bring the RCCL communicator up HERE, as the reference does -- a process launched as one of WORLD_SIZE ranks never
        # trains without it (the library refuses the step, csrc/dist.cpp::dist_require_ready)
        self.rank, self.world, self.local_rank = parallel.rank_world_from_env()
        launcher_rank = self.rank
        if self.world > 1 and not parallel.allow_unsynced():
            r, w = parallel.init_from_env()
            if (r, w) != (self.rank, self.world):
                raise RuntimeError(f'RCCL communicator is rank {r}/{w}, launcher says {self.rank}/{self.world}')
        elif self.world > 1:
            # independent replicas on purpose (DL4DS_ALLOW_UNSYNCED): train as a single process (no collectives, no LR
            # scaling, the whole data set) -- but only the launcher's rank 0 writes result files, so the replicas do not
            # overwrite each other's checkpoints / best_model / loss histories in one save_path
            self.rank, self.world = 0, 1
        n_devices = 1           # per process; the reference's list_physical_devices quirk (base.py:108-116) is not kept
        self.global_batch_size = self.batch_size * n_devices
        self.running_on_first_worker = launcher_rank == 0
        imsize = self.patch_size if self.patch_size is not None else self.data_train.shape[-2]
        if self.scale is not None:
            if imsize % self.scale != 0:
                raise ValueError('The image size must be divisible by `scale` (remainder must be zero). '
                                 'Crop the images or set `patch_size` accordingly')
            if self.data_train_lr is not None:
                scale_from_data = self.data_train.shape[1] / self.data_train_lr.shape[1]
                if not int(scale_from_data) == int(self.scale):
                    raise ValueError('Wrong `scale` value, check `data_train` and `data_train_lr` grid sizes')
        self.lossf = checkarg_loss(self.loss)

    @abstractmethod
    def run(self):
        pass

    @abstractmethod
    def setup_model(self):
        pass

    def _channels(self, predictors, static_vars):
        """Channel accounting of supervised.py:246-260 / cgan.py:177-193."""
        n_channels = self.data_train.shape[-1]
        n_aux = 0
        if self.model_is_spatiotemporal:
            if predictors is not None:
                n_channels += len(predictors)
            if static_vars is not None:
                n_aux += len(static_vars)
        else:
            if static_vars is not None:
                n_channels += len(static_vars)
                n_aux = len(static_vars)
            if predictors is not None:
                n_channels += len(predictors)
        return n_channels, n_aux

    def _grid_sizes(self):
        if self.patch_size is None:
            hr = (int(self.data_train.shape[1]), int(self.data_train.shape[2]))
            lr = (int(self.data_train.shape[1] / self.scale), int(self.data_train.shape[2] / self.scale))
        else:
            hr = (int(self.patch_size), int(self.patch_size))
            lr = (int(self.patch_size / self.scale),) * 2
        return lr, hr

    def save_results(self, model_to_save=None, folder_prefix=None):
        """base.py:162-187 -- same signature and file layout: the model goes to
        ``save_path + [folder_prefix] + <backbone>_<upsampling>/`` (as named weights, ``model_weights.npz``: TF SavedModel is not
        reproducible without TF), running time and test loss next to it in ``save_path``; first worker only."""
        if not self.save:
            return
        if model_to_save is None:
            model_to_save = self.model
        self.model_save_path = self.save_path + (folder_prefix or '') + self.backbone + '_' + self.upsampling + '/'
        if not self.running_on_first_worker:
            return
        os.makedirs(self.model_save_path, exist_ok=True)
        np.savez(os.path.join(self.model_save_path, 'model_weights.npz'), **model_to_save.get_weights())
        if hasattr(self, 'running_time'):
            np.savetxt(os.path.join(self.save_path, 'running_time.txt'), [self.running_time], fmt='%s')
        if hasattr(self, 'test_loss'):
            np.savetxt(os.path.join(self.save_path, 'test_loss.txt'), [self.test_loss], fmt='%0.6f')
