"""ORACLE (test infrastructure, never imported by dl4ds_amd): CPU restatement of the batch preparation that feeds the
hot path -- dl4ds/dataloader.py:11-360 (create_pair_hr_lr, create_batch_hr_lr) and dl4ds/utils.py:251-401 (crop_array,
resize_array) -- written per sample and per pixel, independently of dl4ds_amd/dataloader.py and csrc/batchprep.hip.

OpenCV is an un-vendored dependency of the reference (setup.py: opencv-python, unpinned) and is absent here, so
``cv2.resize`` is restated from OpenCV's published algorithm (modules/imgproc/src/resize.cpp, 4.x).  With
scale_x = src_w / dst_w (the same along y), for floating-point images:

* INTER_AREA, scale >= 1 integer ("resizeAreaFast_"): dst[y, x] = mean of the scale_y x scale_x source block.
* INTER_AREA, scale > 1 not an integer (computeResizeAreaTab): overlap-weighted mean of the source pixels a destination
  cell covers, separably per axis (``_axis_area_down``); used whenever EITHER axis has a non-integer ratio.
  True area resampling needs scale_x >= 1 AND scale_y >= 1: if either axis grows, BOTH take the next rule.
* INTER_AREA, scale < 1 (up-scaling): OpenCV switches to the bilinear code path with its own coefficients:
  ``sx = floor(dx * scale_x); fx = (dx + 1) - (sx + 1) / scale_x; fx = fx <= 0 ? 0 : fx - floor(fx)``.
  For an INTEGER factor s = 1 / scale_x and dx = sx * s + k (0 <= k < s): fx = k + 1 - s <= 0, so fx = 0 and
  dst[dx] = src[dx // s] -- pixel replication.  (``_inter_area_up`` evaluates the general formula; the identity is a test.)
* INTER_NEAREST: sx = min(floor(dx * scale_x), src_w - 1).
* INTER_LINEAR: fx = (dx + 0.5) * scale_x - 0.5; sx = floor(fx); fx -= sx; sx < 0 -> (0, 0); sx >= src_w - 1 -> (src_w - 1, 0).
* INTER_CUBIC: same centre, taps sx - 1 .. sx + 2 (indices clamped = BORDER_REPLICATE), Keys weights with A = -0.75:
  w0 = ((A (t+1) - 5A)(t+1) + 8A)(t+1) - 4A, w1 = ((A+2) t - (A+3)) t^2 + 1, w2 = ((A+2)(1-t) - (A+3))(1-t)^2 + 1, w3 = 1 - w0 - w1 - w2.
* INTER_LANCZOS4: same centre, taps sx - 3 .. sx + 4 (indices clamped), the a = 4 Lanczos window
  L(t) = sinc(t) sinc(t / 4) evaluated at t = (f + 3 - k), k = 0..7, and normalised to sum 1 (OpenCV's interpolateLanczos4 obtains the
  same eight values from one sin/cos pair and a table of 45-degree rotations; f < FLT_EPSILON -> the centre tap alone).

Parity status: "unpinned" -- nothing here was compared with cv2 output (none is available); the known-answer tests in
tests/test_oracle_dataprep.py pin the formulas to hand-computed values.
"""
import numpy as np

POST = ['spc', 'rc', 'dc']


# ----------------------------------------------------------------------------------------------- cv2.resize restated
def _axis_area_down(n_src, n_dst):
    """INTER_AREA down-scaling along one axis.  Integer ratio ("resizeAreaFast_"): block means.  Any other ratio
    (resize.cpp: computeResizeAreaTab / ResizeArea_): destination cell d covers [d scale, (d + 1) scale) of the source axis
    (scale = n_src / n_dst); every source pixel contributes its OVERLAP with the cell, normalised by the cell width
    min(scale, n_src - d scale) -- with OpenCV's rule that a partial overlap below 1e-3 of a pixel is dropped."""
    W = np.zeros((n_dst, n_src))
    if n_src % n_dst == 0:
        s = n_src // n_dst
        for d in range(n_dst):
            W[d, d * s:(d + 1) * s] = 1.0 / s
        return W
    scale = n_src / n_dst
    for d in range(n_dst):
        f1 = d * scale
        f2 = f1 + scale
        cell = min(scale, n_src - f1)
        s1, s2 = int(np.ceil(f1)), int(np.floor(f2))
        s2 = min(s2, n_src - 1)
        s1 = min(s1, s2)
        if s1 - f1 > 1e-3:
            W[d, s1 - 1] += (s1 - f1) / cell
        for sx in range(s1, s2):
            W[d, sx] += 1.0 / cell
        if f2 - s2 > 1e-3:
            W[d, s2] += min(min(f2 - s2, 1.0), cell) / cell
    return W


def _axis_area_up(n_src, n_dst):
    scale = n_src / n_dst
    W = np.zeros((n_dst, n_src))
    for d in range(n_dst):
        sx = int(np.floor(d * scale))
        fx = (d + 1) - (sx + 1) / scale
        fx = 0.0 if fx <= 0 else fx - np.floor(fx)
        if sx < 0:
            sx, fx = 0, 0.0
        if sx >= n_src - 1:
            sx, fx = n_src - 1, 0.0
        W[d, sx] += 1.0 - fx
        if fx:
            W[d, sx + 1] += fx
    return W


def _axis_nearest(n_src, n_dst):
    scale = n_src / n_dst
    W = np.zeros((n_dst, n_src))
    for d in range(n_dst):
        W[d, min(int(np.floor(d * scale)), n_src - 1)] = 1.0
    return W


def _axis_linear(n_src, n_dst):
    scale = n_src / n_dst
    W = np.zeros((n_dst, n_src))
    for d in range(n_dst):
        f = (d + 0.5) * scale - 0.5
        sx = int(np.floor(f))
        f -= sx
        if sx < 0:
            sx, f = 0, 0.0
        if sx >= n_src - 1:
            sx, f = n_src - 1, 0.0
        W[d, sx] += 1.0 - f
        if f:
            W[d, sx + 1] += f
    return W


def _axis_cubic(n_src, n_dst):
    A = -0.75
    scale = n_src / n_dst
    W = np.zeros((n_dst, n_src))
    for d in range(n_dst):
        f = (d + 0.5) * scale - 0.5
        sx = int(np.floor(f))
        t = f - sx
        w = [((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A,
             ((A + 2) * t - (A + 3)) * t * t + 1,
             ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1]
        w.append(1.0 - w[0] - w[1] - w[2])
        for k in range(4):
            W[d, min(max(sx - 1 + k, 0), n_src - 1)] += w[k]
    return W


def _axis_lanczos4(n_src, n_dst):
    scale = n_src / n_dst
    W = np.zeros((n_dst, n_src))
    for d in range(n_dst):
        f = (d + 0.5) * scale - 0.5
        sx = int(np.floor(f))
        f -= sx
        if f < np.finfo(np.float32).eps:
            w = [0.0] * 8
            w[3] = 1.0
        else:
            w = [float(np.sinc(f + 3 - k) * np.sinc((f + 3 - k) / 4.0)) for k in range(8)]
            tot = sum(w)
            w = [v / tot for v in w]
        for k in range(8):
            W[d, min(max(sx - 3 + k, 0), n_src - 1)] += w[k]
    return W


def cv2_resize(img, size_xy, interpolation):
    """cv2.resize(img, (size_x, size_y), interpolation=...) for a float [y, x(, c)] image."""
    a = np.asarray(img, np.float64)
    squeeze = a.ndim == 2
    if squeeze:
        a = a[..., None]
    size_x, size_y = size_xy
    h, w = a.shape[:2]

    def axis(n_src, n_dst):
        if n_src == n_dst:
            return np.eye(n_src)
        if interpolation == 'inter_area':
            # true area resampling only when NEITHER axis grows (resize.cpp: `scale_x >= 1 && scale_y >= 1`); otherwise BOTH
            # axes run the bilinear path with the "area" coefficients, whose formula holds at any ratio
            return _axis_area_down(n_src, n_dst) if (n_dst < n_src and not any_grows) else _axis_area_up(n_src, n_dst)
        if interpolation == 'nearest':
            return _axis_nearest(n_src, n_dst)
        if interpolation == 'bilinear':
            return _axis_linear(n_src, n_dst)
        if interpolation == 'bicubic':
            return _axis_cubic(n_src, n_dst)
        if interpolation == 'lanczos':
            return _axis_lanczos4(n_src, n_dst)
        raise NotImplementedError(f'cv2 interpolation {interpolation!r} is not restated')
    any_grows = size_y > h or size_x > w
    Wy, Wx = axis(h, size_y), axis(w, size_x)
    out = np.einsum('dy,yxc->dxc', Wy, a)
    out = np.einsum('ex,dxc->dec', Wx, out)
    return out[..., 0] if squeeze else out


def checkarray_ndim(array, ndim=3, add_axis_position=-1):
    """utils.py:46-55."""
    return np.expand_dims(array, axis=add_axis_position) if array.ndim < ndim else array


def resize_array(array, newsize, interpolation='inter_area', squeezed=True):
    """utils.py:330-401: 2-D / 3-D images in one cv2.resize call (a single channel comes back 2-D and is re-expanded), 4-D
    arrays frame by frame."""
    a = np.asarray(array)
    size_x, size_y = newsize
    if a.ndim in (2, 3):
        out = cv2_resize(a, (size_x, size_y), interpolation)
        if out.ndim == 2 and a.ndim == 3:
            out = out[..., None]
    elif a.ndim == 4:
        out = np.zeros((a.shape[0], size_y, size_x, a.shape[-1]))
        for i in range(a.shape[0]):
            out[i] = cv2_resize(a[i], (size_x, size_y), interpolation)
    else:
        raise RuntimeError(f'Wrong dimensions, got {a.ndim}')
    return np.squeeze(out) if squeezed else out


def crop_corner(size_y, size_x, size, randint):
    """utils.py:303-304: ``np.random.randint(0, n - size)`` -- the upper bound is EXCLUSIVE, so the last admissible corner
    n - size is never drawn, and a patch as large as the field raises in numpy; here that case yields corner 0."""
    y = int(randint(0, size_y - size)) if size_y > size else 0
    x = int(randint(0, size_x - size)) if size_x > size else 0
    return y, x


def crop(array, size, y, x):
    """utils.py:306-333 for [y,x(,c)] and [t,y,x,c] arrays."""
    if array.ndim in (2, 3):
        return array[y:y + size, x:x + size]
    return array[:, y:y + size, x:x + size]


# ----------------------------------------------------------------------------------------------- create_pair_hr_lr
def get_season(months, time_window=None):
    """dataloader.py:508-525 on month numbers: the month itself, or -- with a time window -- the COUNT of the most frequent
    month (`int(scipy.stats.mode(months).count)`, dataloader.py:514-515: the reference's own reading)."""
    months = np.atleast_1d(np.asarray(months)).astype(int)
    if time_window is None:
        m = int(months.ravel()[0])
    else:
        best = 0
        for v in set(months.tolist()):
            best = max(best, int((months == v).sum()))
        m = best
    for name, ms in (('winter', (12, 1, 2)), ('spring', (3, 4, 5)), ('summer', (6, 7, 8)), ('autumn', (9, 10, 11))):
        if m in ms:
            return name
    raise ValueError(f'no season for month {m}')


def season_array(season, sizey, sizex):
    """dataloader.py:528-542."""
    order = ['winter', 'spring', 'summer', 'autumn']
    if season not in order:
        raise ValueError('``season`` not recognized')
    a = np.zeros((sizey, sizex, 4))
    a[:, :, order.index(season)] += 1
    return a


def create_pair_hr_lr(array, array_lr, upsampling, scale, patch_size, static_vars=None, predictors=None,
                      interpolation='inter_area', randint=None, season=None):
    """dataloader.py:11-294.  ``randint(lo, hi)`` replaces np.random.randint (hi exclusive).
    Returns (hr, lr[, static_hr], (crop_y, crop_x) in HR pixels or None)."""
    hr = np.asarray(array)
    spt = hr.ndim == 4
    nd = 4 if spt else 3
    hr_y, hr_x = (hr.shape[1], hr.shape[2]) if spt else (hr.shape[0], hr.shape[1])
    lr_given = array_lr is not None
    crop_y = crop_x = None
    if upsampling == 'pin':                                                     # dataloader.py:88-141
        if lr_given:
            lr_in = checkarray_ndim(np.asarray(array_lr), nd)
            lr_y, lr_x = (lr_in.shape[1], lr_in.shape[2]) if spt else (lr_in.shape[0], lr_in.shape[1])
            lr_res = resize_array(lr_in, (hr_x, hr_y), interpolation, squeezed=False)
        else:
            lr_x, lr_y = int(hr_x / scale), int(hr_y / scale)
            lr_res = resize_array(hr, (lr_x, lr_y), interpolation, squeezed=False)
            lr_res = resize_array(lr_res, (hr_x, hr_y), interpolation, squeezed=False)
        hr_out = checkarray_ndim(hr, nd)
        lr = checkarray_ndim(lr_res, nd)
        if patch_size is not None:
            crop_y, crop_x = crop_corner(hr_y, hr_x, patch_size, randint)
            hr_out = crop(hr_out, patch_size, crop_y, crop_x)
            lr = crop(lr, patch_size, crop_y, crop_x)
        if predictors is not None:
            p = np.asarray(predictors)
            if p.shape[-3] != lr_y or p.shape[-2] != lr_x:
                p = resize_array(p, (lr_x, lr_y), interpolation, squeezed=False)
            p = checkarray_ndim(resize_array(p, (hr_x, hr_y), interpolation, squeezed=False), nd)
            if patch_size is not None:
                p = crop(p, patch_size, crop_y, crop_x)
            lr = np.concatenate([lr, p], axis=-1)
    elif upsampling in POST:                                                    # dataloader.py:143-214
        ps_lr = None if patch_size is None else int(patch_size / scale)
        if lr_given:
            lr = checkarray_ndim(np.asarray(array_lr), nd)
            lr_y, lr_x = (lr.shape[1], lr.shape[2]) if spt else (lr.shape[0], lr.shape[1])
        else:
            lr = None
            lr_x, lr_y = int(hr_x / scale), int(hr_y / scale)
        hr_out = checkarray_ndim(hr, nd)
        if predictors is not None:
            p = np.asarray(predictors)
            if p.shape[-3] != lr_y or p.shape[-2] != lr_x:
                p = resize_array(p, (lr_x, lr_y), interpolation, squeezed=False)
            p = checkarray_ndim(p, nd)
            if patch_size is not None:
                # :166-174 -- the crop is drawn on the LR predictors and scaled to the HR grid
                cy_lr, cx_lr = crop_corner(lr_y, lr_x, ps_lr, randint)
                p = crop(p, ps_lr, cy_lr, cx_lr)
                crop_y, crop_x = int(cy_lr * scale), int(cx_lr * scale)
                hr_out = crop(hr_out, patch_size, crop_y, crop_x)
                if lr_given:
                    lr = crop(lr, ps_lr, cy_lr, cx_lr)
            if not lr_given:
                # :177-178 resizes the (cropped) HR array to (lr_x, lr_y) -- the FULL LR size, which cannot be concatenated
                # with patch-sized predictors (the reference raises there); the only consistent reading is the patch's LR size
                tx, ty = (lr_x, lr_y) if patch_size is None else (ps_lr, ps_lr)
                lr = checkarray_ndim(resize_array(hr_out, (tx, ty), interpolation, squeezed=False), nd)
            lr = np.concatenate([lr, p], axis=-1)
        else:
            if patch_size is not None:
                if lr_given:                                                    # :193-200
                    cy_lr, cx_lr = crop_corner(lr_y, lr_x, ps_lr, randint)
                    lr = crop(lr, ps_lr, cy_lr, cx_lr)
                    crop_y, crop_x = int(cy_lr * scale), int(cx_lr * scale)
                    hr_out = crop(hr_out, patch_size, crop_y, crop_x)
                else:                                                           # :201-205: crop HR anywhere, coarsen the PATCH
                    crop_y, crop_x = crop_corner(hr_y, hr_x, patch_size, randint)
                    hr_out = crop(hr_out, patch_size, crop_y, crop_x)
                    lr = checkarray_ndim(resize_array(hr_out, (ps_lr, ps_lr), interpolation, squeezed=False), nd)
            elif not lr_given:
                lr = checkarray_ndim(resize_array(hr_out, (lr_x, lr_y), interpolation, squeezed=False), nd)
    else:
        raise ValueError(f'unknown upsampling {upsampling}')
    static_hr = None
    if static_vars is not None:                                                 # :52-68, :218-226
        stat = []
        for var in static_vars:
            v = checkarray_ndim(np.squeeze(np.asarray(var)), 3)
            if patch_size is not None:
                v = crop(v, patch_size, crop_y, crop_x)
            if upsampling in POST:
                ty, tx = (lr.shape[-3], lr.shape[-2])
                v_lr = checkarray_ndim(resize_array(v, (tx, ty), interpolation, squeezed=False), 3)
            else:
                v_lr = v
            stat.append(v)
            if not spt:
                lr = np.concatenate([lr, v_lr], axis=-1)
        static_hr = np.concatenate(stat, axis=-1).astype('float32')
    if season is not None:                                                      # :224-245
        if static_hr is None:
            raise ValueError('season channels need static_vars (np.concatenate([[], season_array]) fails in the reference)')
        post = upsampling in POST
        if patch_size is not None:
            hr_sz = (patch_size, patch_size)
            lr_sz = (int(patch_size / scale),) * 2 if post else hr_sz
            to_lr = True
        else:
            hr_sz = (hr_y, hr_x)
            lr_sz = (int(hr_y / scale), int(hr_x / scale)) if post else hr_sz
            to_lr = not spt
        static_hr = np.concatenate([static_hr, season_array(season, *hr_sz)], axis=-1).astype('float32')
        if to_lr:
            lr = np.concatenate([lr, season_array(season, *lr_sz)], axis=-1)
    out = [np.asarray(hr_out, 'float32'), np.asarray(lr, 'float32')]
    if static_hr is not None:
        out.append(static_hr)
    out.append(None if crop_y is None else (crop_y, crop_x))
    return tuple(out)


def create_batch_hr_lr(all_indices, index, array, array_lr, upsampling, scale=4, batch_size=32, patch_size=None,
                       time_window=None, static_vars=None, predictors=None, interpolation='inter_area', randint=None):
    """dataloader.py:297-360.  Returns ([lr(, static)], [hr], crops)."""
    idx = all_indices[index * batch_size:(index + 1) * batch_size]
    b_hr, b_lr, b_aux, crops = [], [], [], []
    for i in idx:
        if time_window is None:
            d, dl = array[i], (None if array_lr is None else array_lr[i])
            p = None if predictors is None else predictors[i]
        else:
            d = array[i:i + time_window]
            dl = None if array_lr is None else array_lr[i:i + time_window]
            p = None if predictors is None else predictors[i:i + time_window]
        res = create_pair_hr_lr(d, dl, upsampling, scale, patch_size, static_vars=static_vars, predictors=p,
                                interpolation=interpolation, randint=randint)
        b_hr.append(res[0])
        b_lr.append(res[1])
        if static_vars is not None:
            b_aux.append(res[2])
        crops.append(res[-1])
    if static_vars is not None:
        return [np.asarray(b_lr), np.asarray(b_aux)], [np.asarray(b_hr)], crops
    return [np.asarray(b_lr)], [np.asarray(b_hr)], crops
