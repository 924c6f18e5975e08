"""Argument validation of the reference boundary (dl4ds/utils.py:58-171): same conditions, same
exception types."""
from . import (BACKBONE_BLOCKS, UPSAMPLING_METHODS, DROPOUT_VARIANTS, LOSS_FUNCTIONS)


def check_compatibility_upsbackb(backbone, upsampling, time_window):
    """utils.py:58-80."""
    upsampling = checkarg_upsampling(upsampling)
    backbone = checkarg_backbone(backbone)
    if backbone == 'unet' and upsampling != 'pin':
        raise ValueError('`unet` backbone only works with `pin` pre-upsampling')
    if backbone in ['convnext', 'unet'] and time_window is not None:
        raise ValueError('`unet` and `convnext` backbones only work with spatial samples '
                         '(`time_window` must be None)')
    return backbone, upsampling


def checkarg_upsampling(upsampling):
    """utils.py:83-99."""
    if not isinstance(upsampling, str):
        raise TypeError('`upsampling` must be a string')
    if upsampling not in UPSAMPLING_METHODS:
        raise ValueError(f'`upsampling` not recognized. Must be one of the following: '
                         f'{UPSAMPLING_METHODS}. Got {upsampling}')
    return upsampling


def checkarg_backbone(backbone):
    """utils.py:102-118."""
    if not isinstance(backbone, str):
        raise TypeError('`backbone` must be a string')
    if backbone not in BACKBONE_BLOCKS:
        raise ValueError(f'`backbone` not recognized. Must be one of the following: '
                         f'{BACKBONE_BLOCKS}. Got {backbone}')
    return backbone


def checkarg_dropout_variant(dropout_variant):
    """utils.py:121-136."""
    if dropout_variant is None or dropout_variant == 'vanilla':
        return dropout_variant
    elif isinstance(dropout_variant, str):
        if dropout_variant not in DROPOUT_VARIANTS:
            raise ValueError(f'`dropout_variant` must be None or one of {DROPOUT_VARIANTS}, got {dropout_variant}')
        return dropout_variant


def checkarg_loss(loss):
    """utils.py:139-171 -- returns the loss NAME (the kernels are selected by name)."""
    if isinstance(loss, str):
        if loss not in LOSS_FUNCTIONS:
            raise ValueError(f'`loss` must be one of {LOSS_FUNCTIONS}, got {loss}')
        return loss
    raise TypeError(f'`loss` must be a string, one of {LOSS_FUNCTIONS}')


def spatial_to_spatiotemporal_samples(array, time_window):
    """[n_samples, lat, lon, vars] -> [n_samples - time_window + 1, time_window, lat, lon, vars] (utils.py:20-29)."""
    import numpy as np
    n_samples = array.shape[0]
    n_t = n_samples - (time_window - 1)
    return np.stack([array[i:i + time_window] for i in range(n_t)]).astype(np.float64)


def spatiotemporal_to_spatial_samples(array, time_window):
    """Collapse the time-window axis of [n_samples, time_window, lat, lon, vars] back into a sequence of grids: the first
    frame of every window, then the remaining frames of the last window (utils.py:32-45)."""
    import numpy as np
    if array.shape[1] != time_window:
        raise ValueError('`time_window` must be located in the second position [n_samples, time_window, lat, lon, vars]')
    return np.concatenate([array[:, 0], array[-1, 1:]], axis=0)
