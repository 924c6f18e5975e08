"""dl4ds/losses.py:5-149 on MI355X: callables with the reference signature ``loss(y_true, y_pred) -> float``.
Inside the trainers the fused loss+gradient kernels are selected by NAME; these wrappers exist for
stand-alone evaluation."""
from . import ops as _ops


def _make(kind):
    def f(y_true, y_pred):
        return _ops.loss(kind, y_true, y_pred, want_grad=False)[0]
    f.__name__ = kind
    f.__doc__ = f'{kind} (dl4ds/losses.py) evaluated by the gfx950 loss kernels'
    return f


mae = _make('mae')
mse = _make('mse')
dssim = _make('dssim')
dssim_mae = _make('dssim_mae')
dssim_mse = _make('dssim_mse')
dssim_mae_mse = _make('dssim_mae_mse')
msdssim = _make('msdssim')                       # losses.py:92-130 (tf.image.ssim_multiscale, four scales)
msdssim_mae = _make('msdssim_mae')               # losses.py:133-139
msdssim_mae_mse = _make('msdssim_mae_mse')       # losses.py:142-149

__all__ = ['mae', 'mse', 'dssim', 'dssim_mae', 'dssim_mse', 'dssim_mae_mse', 'msdssim', 'msdssim_mae', 'msdssim_mae_mse']
