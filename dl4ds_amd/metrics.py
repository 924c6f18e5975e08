"""compute_metrics of dl4ds/metrics.py:102-330 without the plotting: the per-pair and per-grid-point test metrics are
reduced on the device (`dl4ds_metrics`), the summary statistics the reference prints are assembled here.  Spearman rank
correlations are not computed (they need a sort per grid point / per pair; scipy on the returned arrays does that)."""
import ctypes

import numpy as np

from . import _lib
from .device import DeviceArray
from .dataloader import checkarray_ndim


def image_metrics(y_test, y_test_hat):
    """Raw device reductions -> dict of arrays: per pair ``mae``, ``mse``, ``rmse``, ``psnr``, ``ssim``, ``pearson``; per grid
    point ``rmse_map``, ``bias_map``, ``pearson_map``; ``drange``."""
    y = np.ascontiguousarray(y_test, np.float32)
    p = np.ascontiguousarray(y_test_hat, np.float32)
    if y.shape != p.shape or y.ndim != 4:
        raise ValueError(f'expected two (N, H, W, C) arrays of one shape, got {y.shape} and {p.shape}')
    n, h, w, c = y.shape
    dy, dp = DeviceArray.from_numpy(y), DeviceArray.from_numpy(p)
    pair, grid, rng = DeviceArray.zeros((n, 4)), DeviceArray.zeros((3, h, w, c)), DeviceArray.zeros((2,))
    _lib.check(_lib.lib().dl4ds_metrics(dy.ptr, dp.ptr, n, h, w, c, pair.ptr, grid.ptr, rng.ptr))
    pair, grid, rng = pair.numpy().astype(np.float64), grid.numpy(), rng.numpy().astype(np.float64)
    drange = float(rng[1] - rng[0])
    mse = pair[:, 1]
    with np.errstate(divide='ignore'):
        psnr = 20.0 * np.log10(drange) - 10.0 * np.log10(mse)          # tf.image.psnr(y, y_hat, max_val=drange)
    return dict(mae=pair[:, 0], mse=mse, rmse=np.sqrt(mse), psnr=psnr, ssim=pair[:, 3], pearson=pair[:, 2],
                rmse_map=grid[0], bias_map=grid[1], pearson_map=grid[2], drange=drange)


def compute_metrics(y_test, y_test_hat, dpi=150, plot_size_px=1000, n_jobs=-1, scaler=None, mask=None, save_path=None,
                    verbose=True):
    """Same preparation as the reference (squeeze 5-D, optional ``scaler.inverse_transform``, optional validity mask) and the
    same printed summary; returns ``(temp_rmse_map, temp_pearson_corrmap, nmeanbias)`` like metrics.py:326, plus the full
    dictionary of per-pair / per-grid-point arrays as a fourth element."""
    y_test, y_test_hat = np.asarray(y_test), np.asarray(y_test_hat)
    if y_test.ndim == 5:
        y_test, y_test_hat = np.squeeze(y_test, -1), np.squeeze(y_test_hat, -1)
    y_test, y_test_hat = checkarray_ndim(y_test, 4, -1), checkarray_ndim(y_test_hat, 4, -1)
    if scaler is not None and hasattr(scaler, 'inverse_transform'):
        y_test, y_test_hat = scaler.inverse_transform(y_test), scaler.inverse_transform(y_test_hat)
    mask_nan = None
    if mask is not None:
        mask = np.asarray(getattr(mask, 'values', mask)).copy()
        if mask.ndim == 2:
            mask = mask[..., None]
        y_test, y_test_hat = y_test * mask, y_test_hat * mask
        mask_nan = np.where(mask == 0, np.nan, 1.0)
    m = image_metrics(y_test, y_test_hat)
    rmse_map, corr_map = m['rmse_map'].astype(np.float64), m['pearson_map'].astype(np.float64)
    nmeanbias = m['bias_map'].astype(np.float64) / (np.mean(y_test) * 100)            # metrics.py:219-220
    norm_rmse_map = rmse_map / (np.mean(y_test) * 100)
    if mask_nan is not None:
        rmse_map, corr_map, nmeanbias, norm_rmse_map = (a * mask_nan for a in (rmse_map, corr_map, nmeanbias, norm_rmse_map))
    summary = {
        'PSNR': (np.mean(m['psnr']), np.std(m['psnr'])), 'SSIM': (np.mean(m['ssim']), np.std(m['ssim'])),
        'MAE': (np.mean(m['mae']), np.std(m['mae'])),
        'Per-grid-point RMSE': (np.nanmean(rmse_map), np.nanstd(rmse_map)),
        'Per-grid-point nRMSE': (np.nanmean(norm_rmse_map), np.nanstd(norm_rmse_map)),
        'Per-grid-point Pearson correlation': (np.nanmean(corr_map), np.nanstd(corr_map)),
        'Spatial MSE': (np.mean(m['rmse']), np.std(m['rmse'])),
        'Spatial Pearson correlation': (np.mean(m['pearson']), np.std(m['pearson'])),
    }
    m['summary'] = summary
    if verbose or save_path is not None:
        lines = ['Metrics on y_test and y_test_hat:\n'] + [f'{k} \tmu = {a} \tsigma = {b}' for k, (a, b) in summary.items()]
        if save_path is not None:
            import os
            with open(os.path.join(save_path, 'metrics_summary.txt'), 'a') as f:
                f.write('\n'.join(lines) + '\n')
        elif verbose:
            print('\n'.join(lines))
    if mask is not None:
        for a in (rmse_map, corr_map, nmeanbias):
            a[np.where(np.broadcast_to(mask, a.shape) == 0)] = 0
    return rmse_map, corr_map, nmeanbias, m
