"""Predictor / predict -- dl4ds/inference.py:12-255: forward-only entry on the same graph runtime."""
import numpy as np

from . import POSTUPSAMPLING_METHODS
from .dataloader import create_batch_hr_lr


def predict(trainer, array, scale, array_in_hr=True, static_vars=None, predictors=None, time_window=None,
            time_metadata=None, interpolation='inter_area', batch_size=64, scaler=None, save_path=None,
            save_fname='y_hat.npy', return_lr=False, device='GPU'):
    """inference.py:109-255, step for step: HR arrays are coarsened by the batch builder, LR arrays are first re-expanded
    to the HR grid (``resize_array``, :196-199) and handed over as ``array_lr``; one batch of all
    ``n - (time_window - 1)`` samples (:186-189) goes through ``model.predict``."""
    from .dataloader import resize_array, checkarray_ndim
    if hasattr(trainer, 'model'):
        model = trainer.model
    elif hasattr(trainer, 'generator'):
        model = trainer.generator
    else:
        model = trainer
    upsampling = model.name.split('_')[-1]                       # inference.py:172
    if len(model.input_shapes[0]) == 4 and time_window is None:  # (T,H,W,C) per sample == a 5-D Keras input, :173-175
        raise ValueError('`time_window` must be provided for spatiotemporal model')
    array = np.asarray(getattr(array, 'values', array))
    if static_vars is not None:
        static_vars = [np.asarray(getattr(v, 'values', v)) for v in static_vars]
    n_samples = array.shape[0]
    if time_window is not None:
        n_samples -= time_window - 1                             # inference.py:187-189
    preds = None if predictors is None else np.concatenate([np.asarray(p) for p in predictors], axis=-1)
    if array_in_hr:
        array_hr, array_lr = array, None
    else:
        array = checkarray_ndim(array, 4, -1)
        hr_xy = (array.shape[2] * scale, array.shape[1] * scale)
        array_hr = resize_array(array, hr_xy, interpolation, squeezed=False)
        array_lr = array
    x, _ = create_batch_hr_lr(np.arange(n_samples), 0, array_hr, array_lr, upsampling=upsampling, scale=scale,
                              batch_size=n_samples, patch_size=None, time_window=time_window, static_vars=static_vars,
                              predictors=preds, interpolation=interpolation)
    inputs = x
    y = model.predict(inputs, batch_size=batch_size, verbose=0)
    if y.ndim == 5 and time_window is not None:                  # inference.py:241-242
        from .utils import spatiotemporal_to_spatial_samples
        y = spatiotemporal_to_spatial_samples(y, time_window)
    if scaler is not None:
        y = scaler.inverse_transform(y)
    y = np.asarray(y, np.float32)
    if save_path is not None and save_fname is not None:
        np.save(save_path + ('' if save_path.endswith('/') else '/') + save_fname, y)
    return (y, np.asarray(inputs[0])) if return_lr else y


class Predictor:
    """inference.py:12-106."""

    def __init__(self, trainer, array, scale, array_in_hr=False, static_vars=None, predictors=None, time_window=None,
                 time_metadata=None, interpolation='inter_area', batch_size=64, scaler=None, save_path=None, save_fname='y_hat.npy',
                 return_lr=False, device='GPU'):
        self.kw = dict(trainer=trainer, array=array, scale=scale, array_in_hr=array_in_hr, static_vars=static_vars,
                       predictors=predictors, time_window=time_window, time_metadata=time_metadata,
                       interpolation=interpolation,
                       batch_size=batch_size, scaler=scaler, save_path=save_path, save_fname=save_fname,
                       return_lr=return_lr, device=device)

    def run(self):
        return predict(**self.kw)
