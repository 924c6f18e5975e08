"""Predictor / predict -- dl4ds/inference.py:12-255: forward-only entry on the same graph runtime."""
import numpy as np

from . import POSTUPSAMPLING_METHODS
from .dataloader import create_batch_hr_lr


def predict(trainer, array, scale, array_in_hr=False, static_vars=None, predictors=None, time_window=None,
            interpolation='inter_area', batch_size=64, scaler=None, save_path=None, save_fname='y_hat.npy',
            return_lr=False, device='GPU'):
    """inference.py:109-255."""
    model = getattr(trainer, 'model', None) or getattr(trainer, 'generator', None) or trainer
    upsampling = model.name.split('_')[-1]                       # inference.py:172
    array = np.asarray(getattr(array, 'values', array))
    if static_vars is not None:
        static_vars = [np.asarray(getattr(v, 'values', v)) for v in static_vars]
    preds = None if predictors is None else np.concatenate([np.asarray(p) for p in predictors], axis=-1)
    n = array.shape[0] - (time_window or 0)
    idx = np.arange(n)
    if array_in_hr:
        hr, lr = array, None
    else:
        # the LR array is the model input: for post-upsampling models it is used as is
        hr, lr = None, array
    if hr is None and upsampling in POSTUPSAMPLING_METHODS:
        # build inputs directly from the LR array (+ coarsened statics)
        xs = lr if time_window is None else np.stack([lr[i:i + time_window] for i in idx])
        inputs = [np.asarray(xs, np.float32)]
        if preds is not None:
            ps = preds if time_window is None else np.stack([preds[i:i + time_window] for i in idx])
            inputs[0] = np.concatenate([inputs[0], ps.astype(np.float32)], axis=-1)
        if static_vars is not None:
            from .dataloader import resize_array, checkarray_ndim
            lr_h, lr_w = lr.shape[-3], lr.shape[-2]
            st_hr = np.concatenate([checkarray_ndim(np.squeeze(v), 3) for v in static_vars], axis=-1).astype(np.float32)
            if time_window is None:
                st_lr = np.concatenate([checkarray_ndim(resize_array(checkarray_ndim(np.squeeze(v), 3), (lr_w, lr_h),
                                                                     interpolation, squeezed=False), 3)
                                        for v in static_vars], axis=-1).astype(np.float32)
                inputs[0] = np.concatenate([inputs[0], np.broadcast_to(st_lr, inputs[0].shape[:1] + st_lr.shape)], axis=-1)
            inputs.append(np.broadcast_to(st_hr, inputs[0].shape[:1] + st_hr.shape).copy())
    else:
        src = hr if hr is not None else lr
        x, _ = create_batch_hr_lr(idx, 0, src, None if hr is not None else lr, upsampling=upsampling, scale=scale,
                                  batch_size=n, patch_size=None, time_window=time_window, static_vars=static_vars,
                                  predictors=preds, interpolation=interpolation)
        inputs = x
    y = model.predict(inputs, batch_size=batch_size, verbose=0)
    if y.ndim == 5 and time_window is not None:                  # inference.py:241-242
        from .utils import spatiotemporal_to_spatial_samples
        y = spatiotemporal_to_spatial_samples(y, time_window)
    if scaler is not None:
        y = scaler.inverse_transform(y)
    y = np.asarray(y, np.float32)
    if save_path is not None:
        np.save(save_path + ('' if save_path.endswith('/') else '/') + save_fname, y)
    return (y, inputs[0]) if return_lr else y


class Predictor:
    """inference.py:12-106."""

    def __init__(self, trainer, array, scale, array_in_hr=False, static_vars=None, predictors=None, time_window=None,
                 interpolation='inter_area', batch_size=64, scaler=None, save_path=None, save_fname='y_hat.npy',
                 return_lr=False, device='GPU'):
        self.kw = dict(trainer=trainer, array=array, scale=scale, array_in_hr=array_in_hr, static_vars=static_vars,
                       predictors=predictors, time_window=time_window, interpolation=interpolation,
                       batch_size=batch_size, scaler=scaler, save_path=save_path, save_fname=save_fname,
                       return_lr=return_lr, device=device)

    def run(self):
        return predict(**self.kw)
