"""Host-side batch preparation (numpy) -- the step just BEFORE the hot path.

Mirrors dl4ds/dataloader.py:11-505 and dl4ds/utils.py:251-401 for the cases the trainers use: random
square crops, coarsening by `scale` (cv2.INTER_AREA at an integer ratio == block mean), re-expansion for
'pin' models, predictor / static-variable channel stacking, spatio-temporal windows.  OpenCV is not a
dependency here: ``cv2.resize`` is evaluated from OpenCV's published formulas (resize.cpp) as separable
gathers -- INTER_AREA (integer-ratio block mean; up-scaling through OpenCV's own bilinear-path coefficients,
which replicate pixels at integer factors), INTER_NEAREST, INTER_LINEAR (half-pixel centres, edge clamp),
INTER_CUBIC (A = -0.75, replicated border) and INTER_LANCZOS4 (8 taps, replicated border).  oracle/dataprep.py restates the same
formulas independently (dense per-pixel form) and tests/test_oracle_dataprep.py compares the two.
Season/time-metadata channels are not implemented.
"""
import numpy as np

from . import POSTUPSAMPLING_METHODS, INTERPOLATION_METHODS


def equal_shard(perm, rank, world):
    """perm[rank::world] after dropping the n % world remainder (same rule as dl4ds_amd.parallel.equal_shard; repeated
    here so the host loader does not import the GPU binding)."""
    if world <= 1:
        return perm
    n = (len(perm) // world) * world
    return perm[:n][rank::world]


def checkarray_ndim(array, ndim=3, add_axis_position=-1):
    """utils.py:46-55."""
    if array.ndim < ndim:
        return np.expand_dims(array, axis=add_axis_position)
    return array


def random_corner(sy, sx, size, rng=None):
    """utils.py:303-304 draws ``np.random.randint(0, n - size)``: the upper bound is exclusive, so the last admissible
    corner is never used (and a patch as large as the field raises there; here it gets corner 0).  y first, then x."""
    rng = np.random if rng is None else rng
    draw = (lambda hi: int(rng.randint(0, hi))) if hasattr(rng, 'randint') else (lambda hi: int(rng.integers(0, hi)))
    y = draw(sy - size) if sy > size else 0
    x = draw(sx - size) if sx > size else 0
    return y, x


def crop_array(array, size, yx=None, position=False, rng=None):
    """utils.py:251-327: square crop of a [y,x(,c)] or [t,y,x,c] array; random corner when yx is None."""
    if array.ndim not in [2, 3, 4, 5]:
        raise TypeError('Input array is not a 2D, 3D, or 4D ndarray')
    if not isinstance(size, (int, np.integer)):
        raise TypeError('`Size` must be integer')
    ax = {2: 0, 3: 0, 4: 1, 5: 2}[array.ndim]
    sy, sx = array.shape[ax], array.shape[ax + 1]
    if size > sy or size > sx:
        raise ValueError('`Size` larger than the input image size')
    if yx is not None:
        y, x = yx
    else:
        y, x = random_corner(sy, sx, size, rng)
    sl = [slice(None)] * array.ndim
    sl[ax], sl[ax + 1] = slice(y, y + size), slice(x, x + size)
    out = array[tuple(sl)]
    return (out, y, x) if position else out


def _axis_taps(n_src, n_dst, interpolation, area_linear=False):
    """(indices [n_dst, k], weights [n_dst, k]) of cv2.resize along one axis (OpenCV resize.cpp).  ``area_linear``: INTER_AREA
    while the OTHER axis grows -- OpenCV runs true area resampling only when neither axis grows, otherwise both axes take the
    bilinear code path with its "area" coefficients (the up-scaling branch below, valid at any ratio)."""
    d = np.arange(n_dst)
    scale = n_src / n_dst
    if interpolation == 'inter_area' and n_dst < n_src and not area_linear:
        if n_src % n_dst == 0:
            s = n_src // n_dst
            return d[:, None] * s + np.arange(s)[None, :], np.full((n_dst, s), 1.0 / s)
        # non-integer ratio (cv2 computeResizeAreaTab): destination cell d = [d scale, (d + 1) scale) of the source axis; source
        # pixel j weighs in with its overlap |[j, j + 1) & cell| / cell width, overlaps of the two partial end pixels below 1e-3
        # dropped as OpenCV does.  K = the widest cell's pixel count; unused taps carry weight 0 at a valid index.
        f1 = d * scale
        f2 = f1 + scale
        cell = np.minimum(scale, n_src - f1)
        s2 = np.minimum(np.floor(f2).astype(int), n_src - 1)
        s1 = np.minimum(np.ceil(f1).astype(int), s2)
        K = int((s2 - s1).max()) + 2
        j = (s1 - 1)[:, None] + np.arange(K)[None, :]                 # candidate pixels s1 - 1 ... s1 + K - 2
        head = (j == (s1 - 1)[:, None]) & ((s1 - f1) > 1e-3)[:, None]
        body = (j >= s1[:, None]) & (j < s2[:, None])
        tail = (j == s2[:, None]) & ((f2 - s2) > 1e-3)[:, None]
        w = (head * (s1 - f1)[:, None] + body * 1.0 + tail * np.minimum(np.minimum(f2 - s2, 1.0), cell)[:, None]) / cell[:, None]
        return np.clip(j, 0, n_src - 1), w
    if interpolation == 'nearest':
        return np.minimum(np.floor(d * scale).astype(int), n_src - 1)[:, None], np.ones((n_dst, 1))
    if interpolation in ('inter_area', 'bilinear'):
        if interpolation == 'inter_area':        # up-scaling: OpenCV's "area" coefficients on the bilinear path
            sx = np.floor(d * scale).astype(int)
            f = (d + 1) - (sx + 1) / scale
            f = np.where(f <= 0, 0.0, f - np.floor(f))
        else:
            c = (d + 0.5) * scale - 0.5
            sx = np.floor(c).astype(int)
            f = c - sx
        lo = sx < 0
        hi = sx >= n_src - 1
        sx = np.where(lo, 0, np.where(hi, n_src - 1, sx))
        f = np.where(lo | hi, 0.0, f)
        return np.stack([sx, np.minimum(sx + 1, n_src - 1)], 1), np.stack([1.0 - f, f], 1)
    if interpolation == 'bicubic':
        A = -0.75
        c = (d + 0.5) * scale - 0.5
        sx = np.floor(c).astype(int)
        t = c - sx
        w0 = ((A * (t + 1) - 5 * A) * (t + 1) + 8 * A) * (t + 1) - 4 * A
        w1 = ((A + 2) * t - (A + 3)) * t * t + 1
        w2 = ((A + 2) * (1 - t) - (A + 3)) * (1 - t) * (1 - t) + 1
        idx = np.clip(sx[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, np.stack([w0, w1, w2, 1.0 - w0 - w1 - w2], 1)
    if interpolation == 'lanczos':               # cv2.INTER_LANCZOS4 (interpolateLanczos4: one sin/cos pair, rotated by 45 degrees per tap)
        c = (d + 0.5) * scale - 0.5
        sx = np.floor(c).astype(int)
        t = c - sx
        s45 = 0.70710678118654752440084436210485
        rot = np.array([[1, 0], [-s45, -s45], [0, 1], [s45, -s45], [-1, 0], [s45, s45], [0, -1], [-s45, s45]])
        y0 = -(t + 3) * np.pi * 0.25
        k = np.arange(8)
        y = -(t[:, None] + 3 - k[None, :]) * np.pi * 0.25
        with np.errstate(divide='ignore', invalid='ignore'):
            w = (rot[None, :, 0] * np.sin(y0)[:, None] + rot[None, :, 1] * np.cos(y0)[:, None]) / (y * y)
            w = w / w.sum(1, keepdims=True)
        centre = np.zeros(8)
        centre[3] = 1.0
        w = np.where((t < np.finfo(np.float32).eps)[:, None], centre[None, :], w)
        idx = np.clip(sx[:, None] + np.arange(-3, 5)[None, :], 0, n_src - 1)
        return idx, w
    raise ValueError(f"unknown interpolation '{interpolation}'")


def _resize2d(a, size_y, size_x, interpolation):
    """cv2.resize of a [y,x,c] float array to (size_y, size_x), separable gathers."""
    h, w = a.shape[:2]
    lin = interpolation == 'inter_area' and (size_y > h or size_x > w)
    if h != size_y:
        idx, wt = _axis_taps(h, size_y, interpolation, lin)
        a = np.einsum('dk,dkxc->dxc', wt, a[idx])
    if w != size_x:
        idx, wt = _axis_taps(w, size_x, interpolation, lin)
        a = np.einsum('ek,yekc->yec', wt, a[:, idx])
    return a


def resize_array(array, newsize, interpolation='inter_area', squeezed=True, keep_dynamic_range=False):
    """utils.py:330-401 (newsize is (x, y) like cv2)."""
    if interpolation not in INTERPOLATION_METHODS:
        raise ValueError(f'`interpolation` must be one of {INTERPOLATION_METHODS}. Received {interpolation}')
    size_x, size_y = newsize
    a = np.asarray(array, dtype=np.float64)
    if a.ndim == 2:
        out = _resize2d(a[..., None], size_y, size_x, interpolation)[..., 0]
    elif a.ndim == 3:
        out = _resize2d(a, size_y, size_x, interpolation)
    elif a.ndim == 4:
        out = np.stack([_resize2d(a[i], size_y, size_x, interpolation) for i in range(a.shape[0])])
    else:
        raise RuntimeError(f'Wrong dimensions, got {a.ndim}')
    if squeezed:
        out = np.squeeze(out)
    if keep_dynamic_range:
        out = np.clip(out, a_min=a.min(), a_max=a.max())
    return out


SEASONS = ('winter', 'spring', 'summer', 'autumn')


def get_season(time_metadata, time_window=None):
    """dataloader.py:508-525 (`_get_season_`).  ``time_metadata``: the sample's time stamp(s) -- numpy datetime64 /
    pandas / xarray time values, or plain month numbers.  With a time window the reference takes
    ``int(scipy.stats.mode(months).count)``, i.e. the NUMBER of occurrences of the most frequent month, not that month
    (dataloader.py:514-515); that is what a model trained with the reference has seen, so it is kept."""
    t = getattr(time_metadata, 'values', time_metadata)
    t = np.atleast_1d(np.asarray(getattr(t, 'time', t)))
    if np.issubdtype(t.dtype, np.datetime64):
        months = t.astype('datetime64[M]').astype(int) % 12 + 1
    else:
        months = t.astype(int)
    if time_window is None:
        month = int(months.ravel()[0])
    else:
        _, counts = np.unique(months, return_counts=True)
        month = int(counts.max())
    if month in (12, 1, 2):
        return 'winter'
    if month in (3, 4, 5):
        return 'spring'
    if month in (6, 7, 8):
        return 'summer'
    if month in (9, 10, 11):
        return 'autumn'
    raise ValueError(f'no season for month {month}')       # (the reference leaves `season` unbound here)


def get_season_array(season, sizey, sizex):
    """dataloader.py:528-542: four one-hot channels (winter, spring, summer, autumn)."""
    if season not in SEASONS:
        raise ValueError('``season`` not recognized')
    a = np.zeros((sizey, sizex, 4))
    a[:, :, SEASONS.index(season)] += 1
    return a


def create_pair_hr_lr(array, array_lr, upsampling, scale, patch_size, static_vars=None, predictors=None,
                      season=None, debug=False, interpolation='inter_area', rng=None):
    """dataloader.py:11-294.  Returns (hr, lr[, static_hr]).  ``season``: four one-hot channels appended to the auxiliary
    HR array and (spatial samples) to the LR array (dataloader.py:224-245); like the reference this needs ``static_vars``
    (its `np.concatenate([[], season_array])` fails without them) and, with ``patch_size``, spatial samples."""
    hr = np.asarray(array)
    spt = hr.ndim == 4
    hr_y, hr_x = (hr.shape[1], hr.shape[2]) if spt else (hr.shape[0], hr.shape[1])
    lr_given = array_lr is not None
    nd = 4 if spt else 3
    crop = None
    if upsampling == 'pin':
        if lr_given:
            lr_full = resize_array(checkarray_ndim(np.asarray(array_lr), nd), (hr_x, hr_y), interpolation, squeezed=False)
        else:
            lr_x, lr_y = int(hr_x / scale), int(hr_y / scale)
            lr_full = resize_array(hr, (lr_x, lr_y), interpolation, squeezed=False)
            lr_full = resize_array(lr_full, (hr_x, hr_y), interpolation, squeezed=False)
        hr = checkarray_ndim(hr, nd)
        lr = checkarray_ndim(lr_full, nd)
        if predictors is not None:
            p = resize_array(predictors, (int(hr_x / scale), int(hr_y / scale)), interpolation, squeezed=False)
            p = resize_array(p, (hr_x, hr_y), interpolation, squeezed=False)
            lr = np.concatenate([lr, checkarray_ndim(p, nd)], axis=-1)
        if patch_size is not None:
            hr, cy, cx = crop_array(hr, patch_size, position=True, rng=rng)
            lr = crop_array(lr, patch_size, yx=(cy, cx))
            crop = (cy, cx)
    elif upsampling in POSTUPSAMPLING_METHODS:
        hr = checkarray_ndim(hr, nd)
        ps_lr = None if patch_size is None else int(patch_size / scale)
        if lr_given:
            lr = checkarray_ndim(np.asarray(array_lr), nd)
            lr_y, lr_x = (lr.shape[1], lr.shape[2]) if spt else (lr.shape[0], lr.shape[1])
        else:
            lr = None
            lr_x, lr_y = int(hr_x / scale), int(hr_y / scale)
        p = None
        if predictors is not None:
            p = np.asarray(predictors)
            if p.shape[-3] != lr_y or p.shape[-2] != lr_x:
                p = resize_array(p, (lr_x, lr_y), interpolation, squeezed=False)
            p = checkarray_ndim(p, nd)
        if patch_size is not None:
            if lr_given or p is not None:
                # the crop is drawn on the LR grid and scaled to the HR grid (dataloader.py:166-174,193-200)
                cy_lr, cx_lr = random_corner(lr_y, lr_x, ps_lr, rng)
                crop = (int(cy_lr * scale), int(cx_lr * scale))
                hr = crop_array(hr, patch_size, yx=crop)
                if lr_given:
                    lr = crop_array(lr, ps_lr, yx=(cy_lr, cx_lr))
                if p is not None:
                    p = crop_array(p, ps_lr, yx=(cy_lr, cx_lr))
            else:
                # the HR field is cropped at ANY pixel and the PATCH is coarsened (dataloader.py:201-205)
                hr, cy, cx = crop_array(hr, patch_size, position=True, rng=rng)
                crop = (cy, cx)
        if lr is None:
            # (with predictors the reference resizes the cropped HR array to the FULL LR size, dataloader.py:177-178, and
            # then fails to concatenate; the patch's own LR size is the only consistent reading)
            ty, tx = (lr_y, lr_x) if patch_size is None else (ps_lr, ps_lr)
            lr = checkarray_ndim(resize_array(hr, (tx, ty), interpolation, squeezed=False), nd)
        if p is not None:
            lr = np.concatenate([lr, p], axis=-1)
    else:
        raise ValueError(f'unknown upsampling {upsampling}')
    static_hr = None
    if static_vars is not None:
        stat = []
        for var in static_vars:
            v = checkarray_ndim(np.squeeze(np.asarray(var)), 3)
            if crop is not None:
                v = crop_array(v, patch_size, yx=crop)
            stat.append(v)
            if not spt:
                if upsampling in POSTUPSAMPLING_METHODS:
                    v_lr = checkarray_ndim(resize_array(v, (lr.shape[1], lr.shape[0]), interpolation, squeezed=False), 3)
                else:
                    v_lr = v
                lr = np.concatenate([lr, v_lr], axis=-1)
        static_hr = np.concatenate(stat, axis=-1).astype('float32')
    if season is not None:
        if static_hr is None:
            raise ValueError('season channels need `static_vars` (dataloader.py:224-235 concatenates them to the static array)')
        if patch_size is not None:
            sy = sx = patch_size
            ly = lx = int(patch_size / scale) if upsampling in POSTUPSAMPLING_METHODS else patch_size
            to_lr = True                                   # (dataloader.py:232: also for spatio-temporal samples, where it fails)
        else:
            sy, sx = hr_y, hr_x
            ly, lx = (int(hr_y / scale), int(hr_x / scale)) if upsampling in POSTUPSAMPLING_METHODS else (hr_y, hr_x)
            to_lr = not spt
        static_hr = np.concatenate([static_hr, get_season_array(season, sy, sx)], axis=-1).astype('float32')
        if to_lr:
            lr = np.concatenate([lr, get_season_array(season, ly, lx)], axis=-1)
    hr = np.asarray(hr, 'float32')
    lr = np.asarray(lr, 'float32')
    if static_hr is not None:
        return hr, lr, static_hr
    return hr, lr


def create_batch_hr_lr(all_indices, index, array, array_lr, upsampling, scale=4, batch_size=32, patch_size=None,
                       time_window=None, static_vars=None, predictors=None, interpolation='inter_area',
                       time_metadata=None, rng=None):
    """dataloader.py:297-360."""
    idx = all_indices[index * batch_size:(index + 1) * batch_size]
    b_hr, b_lr, b_aux = [], [], []
    for i in idx:
        if time_window is None:
            d, dl = array[i], (None if array_lr is None else array_lr[i])
            p = None if predictors is None else predictors[i]
        else:
            d, dl = array[i:i + time_window], (None if array_lr is None else array_lr[i:i + time_window])
            p = None if predictors is None else predictors[i:i + time_window]
        season = None
        if time_metadata is not None:                       # dataloader.py:327,334
            season = get_season(time_metadata[i] if time_window is None else time_metadata[i:i + time_window], time_window)
        res = create_pair_hr_lr(d, dl, upsampling, scale, patch_size, static_vars=static_vars, predictors=p,
                                season=season, interpolation=interpolation, rng=rng)
        if static_vars is not None:
            b_aux.append(res[2])
        b_hr.append(res[0])
        b_lr.append(res[1])
    if static_vars is not None:
        return [np.asarray(b_lr), np.asarray(b_aux)], [np.asarray(b_hr)]
    return [np.asarray(b_lr)], [np.asarray(b_hr)]


class DataGenerator:
    """dataloader.py:363-505 (a keras.utils.Sequence there; a plain indexable object here)."""

    def __init__(self, array, array_lr, backbone, upsampling, scale, batch_size=32, patch_size=None, time_window=None,
                 static_vars=None, predictors=None, interpolation='inter_area', repeat=None, seed=None, rank=0,
                 world=1, time_metadata=None):
        # time_metadata: per-time-step stamps; switches the four season channels on.  The reference's generator always
        # passes None (`self.time_metadata = array.time.copy()` is commented out, dataloader.py:428-433), so the season
        # channels are only reachable through create_pair_hr_lr / create_batch_hr_lr there; here it is an explicit option.
        self.time_metadata = None if time_metadata is None else np.asarray(getattr(time_metadata, 'values', time_metadata))
        self.array = np.asarray(getattr(array, 'values', array))
        self.array_lr = None if array_lr is None else np.asarray(getattr(array_lr, 'values', array_lr))
        self.batch_size, self.scale, self.upsampling, self.backbone = batch_size, scale, upsampling, backbone
        self.patch_size, self.time_window = patch_size, time_window
        self.static_vars = None if static_vars is None else [np.asarray(getattr(v, 'values', v)) for v in static_vars]
        self.predictors = None if predictors is None else np.concatenate([np.asarray(p) for p in predictors], axis=-1)
        self.interpolation, self.repeat = interpolation, repeat
        self.n = self.array.shape[0] - self.time_window if self.time_window is not None else self.array.shape[0]
        self.rng = np.random.default_rng(seed)
        perm = self.rng.permutation(self.n)
        # rank-strided shard of one seeded permutation, the same length on every rank (unequal step counts would leave
        # ranks waiting in the gradient all-reduce)
        self.indices = equal_shard(perm, rank, world)
        if self.repeat is not None and isinstance(self.repeat, int):
            self.indices = np.hstack([self.indices for _ in range(self.repeat)])
        if patch_size is not None and upsampling in POSTUPSAMPLING_METHODS and patch_size % scale != 0:
            raise ValueError('`patch_size` must be divisible by `scale`')

    def __len__(self):
        n_batches = len(self.indices) // self.batch_size
        return n_batches

    def __getitem__(self, index):
        return create_batch_hr_lr(self.indices, index, self.array, self.array_lr, upsampling=self.upsampling,
                                  scale=self.scale, batch_size=self.batch_size, patch_size=self.patch_size,
                                  time_window=self.time_window, static_vars=self.static_vars,
                                  predictors=self.predictors, interpolation=self.interpolation,
                                  time_metadata=self.time_metadata, rng=self.rng)


class DeviceDataGenerator:
    """DataGenerator whose dataset lives in HBM and whose batches are gathered on the device (csrc/batchprep.hip) -- SURVEY
    section 8 "next" row f1.  Same constructor, same seeded permutation and the same per-sample RNG calls as DataGenerator, so
    `gen[i]` holds exactly the batch `DataGenerator(...)[i]` would build (to fp32 rounding).  Three routes, chosen once:
      * 'inter_area' on HR-grid inputs whose size `scale` divides (the trainers' default): `dl4ds_batch_prepare`, block-mean /
        replication kernels;
      * every other interpolation of `resize_array` on the same inputs (and inter_area with ``taps=True``):
        `dl4ds_batch_prepare_taps` on per-axis tap tables built once from cv2's coefficients (`_axis_taps`);
      * round 5, everything else the reference's create_pair_hr_lr accepts (dataloader.py:72-73,92-96,149-163,193-200;
        utils.py:369-381) -- a caller-supplied LR array, predictors on the LR (or any other) grid, fields whose size `scale` does
        not divide (cv2.INTER_AREA between grids at a non-integer ratio) -- composed here from `dl4ds_batch_gather` passes.

    `gen[i]` returns ([lr(, static_hr)], [hr]) as DeviceArray objects that stay valid until the next `gen[...]` call
    (two rotating output buffers, so the previous batch can still be in flight); `.numpy()` them for inspection.
    """

    def __init__(self, array, array_lr, backbone, upsampling, scale, batch_size=32, patch_size=None, time_window=None,
                 static_vars=None, predictors=None, interpolation='inter_area', repeat=None, seed=None, rank=0,
                 world=1, taps=None):
        from .device import DeviceArray
        if interpolation not in INTERPOLATION_METHODS:
            raise ValueError(f'`interpolation` must be one of {INTERPOLATION_METHODS}. Received {interpolation}')
        self.interpolation = interpolation
        self.taps = (interpolation != 'inter_area') if taps is None else bool(taps)
        a = np.asarray(getattr(array, 'values', array), np.float32)
        if a.ndim == 3:
            a = a[..., None]
        self.N, self.H, self.W, self.C = a.shape
        self.scale, self.batch_size, self.upsampling, self.backbone = int(scale), int(batch_size), upsampling, backbone
        self.pin = upsampling == 'pin'
        if not self.pin and upsampling not in POSTUPSAMPLING_METHODS:
            raise ValueError(f'unknown upsampling {upsampling}')
        self.patch_size, self.time_window = patch_size, time_window
        if patch_size is not None and not self.pin and patch_size % self.scale != 0:
            raise ValueError('`patch_size` must be divisible by `scale`')
        self.T = 1 if time_window is None else int(time_window)
        self.spt = time_window is not None
        self._hr = DeviceArray.from_numpy(a)
        # the LR grid: the caller's LR array's own (dataloader.py:92-96,145-148), else int(H / scale) x int(W / scale)
        self._lr_src, self.CLR = None, self.C
        self.hl, self.wl = int(self.H / self.scale), int(self.W / self.scale)
        if array_lr is not None:
            al = np.asarray(getattr(array_lr, 'values', array_lr), np.float32)
            if al.ndim == 3:
                al = al[..., None]
            if al.shape[0] != self.N:
                raise ValueError('`array_lr` must hold as many time steps as `array`')
            self._lr_src, self.CLR = DeviceArray.from_numpy(al), al.shape[-1]
            self.hl, self.wl = al.shape[1], al.shape[2]
        self._pred, self.P, self.pred_grid = None, 0, (self.H, self.W)
        if predictors is not None:
            ps = [np.asarray(getattr(q, 'values', q), np.float32) for q in predictors]
            ps = [q[..., None] if q.ndim == 3 else q for q in ps]
            p = np.concatenate(ps, axis=-1)
            if p.shape[0] != self.N:
                raise ValueError('`predictors` must hold as many time steps as `array`')
            self._pred, self.P, self.pred_grid = DeviceArray.from_numpy(p), p.shape[-1], (p.shape[1], p.shape[2])
        self._stat, self.S = None, 0
        if static_vars is not None:
            sv = [checkarray_ndim(np.squeeze(np.asarray(getattr(v, 'values', v), np.float32)), 3) for v in static_vars]
            st = np.concatenate(sv, axis=-1)
            if st.shape[:2] != (self.H, self.W):
                # (the reference crops them with the HR corner and feeds them to the model's HR auxiliary input, dataloader.py:52-68)
                raise ValueError('static variables must be on the HR grid')
            self._stat, self.S = DeviceArray.from_numpy(st), st.shape[-1]
        # every crop stays inside its source (the device gathers do not clamp; the reference's numpy slices would hand back short
        # arrays and fail on shapes later): the patch inside the HR field, its LR counterpart inside the LR grid, and -- where the
        # corner is drawn on the LR grid and scaled (dataloader.py:166-174,193-200) -- the LR grid times `scale` inside the HR field
        if patch_size is not None:
            ps = int(patch_size)
            if ps < 1 or ps > self.H or ps > self.W:
                raise ValueError(f'`patch_size` {ps} does not fit the {self.H} x {self.W} HR field')
            if not self.pin:
                if ps // self.scale < 1 or ps // self.scale > min(self.hl, self.wl):
                    raise ValueError(f'the LR patch ({ps // self.scale}) does not fit the {self.hl} x {self.wl} LR grid')
                if (self._lr_src is not None or self.P) and (self.hl * self.scale > self.H or self.wl * self.scale > self.W):
                    raise ValueError(f'the LR grid {self.hl} x {self.wl} times scale {self.scale} exceeds the {self.H} x {self.W} HR field: '
                                     'crop corners drawn on the LR grid would leave it')
        if self.hl < 1 or self.wl < 1:
            raise ValueError('the LR grid is empty')
        # the composed route: anything but HR-grid inputs of a size `scale` divides
        self.general = (self._lr_src is not None or self.pred_grid != (self.H, self.W) or self.H % self.scale != 0
                        or self.W % self.scale != 0)
        self.n = self.N - self.T if self.spt else self.N
        self.rng = np.random.default_rng(seed)
        perm = self.rng.permutation(self.n)
        self.indices = equal_shard(perm, rank, world)
        if repeat is not None and isinstance(repeat, int):
            self.indices = np.hstack([self.indices for _ in range(repeat)])
        self.psy, self.psx = (self.H, self.W) if patch_size is None else (int(patch_size), int(patch_size))
        self.static_in_lr = bool(self.S and not self.spt)
        cl = self.CLR + self.P + (self.S if self.static_in_lr else 0)
        if self.pin:
            oy, ox = self.psy, self.psx
        elif patch_size is None:
            oy, ox = self.hl, self.wl
        else:
            oy, ox = self.psy // self.scale, self.psx // self.scale
        self.lr_out = (oy, ox)
        B, T = self.batch_size, self.T
        lead = (B, T) if self.spt else (B,)
        self._bufs = []
        for _ in range(2):
            lr = DeviceArray(lead + (oy, ox, cl))
            hr = DeviceArray(lead + (self.psy, self.psx, self.C))
            st = DeviceArray((B, self.psy, self.psx, self.S)) if self.S else None
            self._bufs.append((lr, hr, st))
        self._turn = 0
        self._tap_keep, self._tab_cache = [], {}
        if self.general:
            self._plan_general()
        elif self.taps:
            self._build_tap_tables()

    # ---- per-axis cv2.resize tables, resident in HBM ----------------------------------------------------------------------------
    def _axis_pair(self, src_yx, dst_yx):
        """ctypes (dl4ds_tap_axis * 2) {y, x} of cv2.resize from a src_yx grid to a dst_yx grid (cached per pair of grids)."""
        import ctypes
        from .device import DeviceArray

        class TapAxis(ctypes.Structure):
            _fields_ = [('idx', ctypes.c_void_p), ('wt', ctypes.c_void_p), ('k', ctypes.c_int)]
        key = (tuple(src_yx), tuple(dst_yx))
        if key in self._tab_cache:
            return self._tab_cache[key]
        lin = self.interpolation == 'inter_area' and (dst_yx[0] > src_yx[0] or dst_yx[1] > src_yx[1])
        arr = (TapAxis * 2)()
        for i, (n_src, n_dst) in enumerate(zip(src_yx, dst_yx)):
            if n_src == n_dst:
                idx, wt = np.arange(n_dst)[:, None], np.ones((n_dst, 1))
            else:
                idx, wt = _axis_taps(n_src, n_dst, self.interpolation, lin)
            d_idx = DeviceArray.from_numpy(np.ascontiguousarray(idx, np.int32))
            d_wt = DeviceArray.from_numpy(np.ascontiguousarray(wt, np.float32))
            self._tap_keep += [d_idx, d_wt]
            arr[i].idx, arr[i].wt, arr[i].k = d_idx.ptr, d_wt.ptr, idx.shape[1]
        self._tab_cache[key] = arr
        return arr

    def _build_tap_tables(self):
        """The three resizes of create_pair_hr_lr for HR-grid inputs (dl4ds_batch_prepare_taps)."""
        from .device import DeviceArray
        s = self.scale
        hl, wl = self.H // s, self.W // s
        self._dn_patch = self._dn_field = self._up_field = self._scratch = None
        if self.pin:
            self._dn_field = self._axis_pair((self.H, self.W), (hl, wl))
            self._up_field = self._axis_pair((hl, wl), (self.H, self.W))
            self._scratch = DeviceArray((self.batch_size, self.T, hl, wl, self.C + self.P))
        else:
            self._dn_patch = self._axis_pair((self.psy, self.psx), (self.psy // s, self.psx // s))
            if self.P:
                self._dn_field = self._axis_pair((self.H, self.W), (hl, wl))

    # ---- the composed route (round 5) ---------------------------------------------------------------------------------------------
    def _plan_general(self):
        """Gather passes (dl4ds_batch_gather) for the inputs create_pair_hr_lr accepts beyond HR-grid fields of a divisible size.
        Each pass = (groups, crop?, output buffer selector, out_h, out_w, T); a group = dict(src, channels, frames, grid, raw,
        origin_from_crop, row_div, table)."""
        from .device import DeviceArray
        H, W, hl, wl, s = self.H, self.W, self.hl, self.wl, self.scale
        B, T = self.batch_size, self.T
        ident = None
        self._passes = []
        self._scratch_g = None

        def grp(src, channels, frames, grid, raw=0, origin_from_crop=0, row_div=0, table=None):
            return dict(src=src, channels=channels, frames=frames, grid=grid, raw=raw, origin_from_crop=origin_from_crop,
                        row_div=row_div, table=table)
        lr_groups = []
        if self.pin:
            # dataloader.py:88-141: LR part = the caller's LR array, or the HR field coarsened to the LR grid, resized to the HR grid;
            # predictors brought to the LR grid unless they are on it, then to the HR grid; everything cropped on the HR grid
            down = []
            if self._lr_src is None:
                down.append(grp(self._hr, self.C, 0, (H, W), table=self._axis_pair((H, W), (hl, wl))))
            if self.P and self.pred_grid != (hl, wl):
                down.append(grp(self._pred, self.P, 0, self.pred_grid, table=self._axis_pair(self.pred_grid, (hl, wl))))
            n_down = sum(g['channels'] for g in down)
            if down:
                self._scratch_g = DeviceArray((B, T, hl, wl, n_down))
                self._passes.append(dict(groups=down, crop=False, out='scratch', oh=hl, ow=wl, T=T))
            up = self._axis_pair((hl, wl), (H, W))
            pred_direct = self.P and self.pred_grid == (hl, wl)
            if self._lr_src is not None:
                lr_groups.append(grp(self._lr_src, self.CLR, 0, (hl, wl), row_div=1, table=up))
                if self.P:
                    lr_groups.append(grp(self._pred, self.P, 0, (hl, wl), row_div=1, table=up) if pred_direct
                                     else grp(self._scratch_g, self.P, 1, (hl, wl), row_div=1, table=up))
            else:
                if self.P and not pred_direct:
                    lr_groups.append(grp(self._scratch_g, self.C + self.P, 1, (hl, wl), row_div=1, table=up))
                else:
                    lr_groups.append(grp(self._scratch_g, self.C, 1, (hl, wl), row_div=1, table=up))
                    if self.P:
                        lr_groups.append(grp(self._pred, self.P, 0, (hl, wl), row_div=1, table=up))
            if self.static_in_lr:
                lr_groups.append(grp(self._stat, self.S, 2, (H, W), raw=1))
        else:
            # dataloader.py:143-214: LR part = the caller's LR array cropped on its own grid, or the HR crop (the whole field without
            # a patch) coarsened; predictors on the LR grid cropped there, others resized as whole fields first; static variables
            # cropped on the HR grid, then coarsened to the LR patch
            oy, ox = self.lr_out
            if self._lr_src is not None:
                lr_groups.append(grp(self._lr_src, self.CLR, 0, (hl, wl), raw=1, row_div=s))
            else:
                lr_groups.append(grp(self._hr, self.C, 0, (H, W), origin_from_crop=1,
                                     table=self._axis_pair((self.psy, self.psx), (oy, ox))))
            if self.P:
                if self.pred_grid == (hl, wl):
                    lr_groups.append(grp(self._pred, self.P, 0, (hl, wl), raw=1, row_div=s))
                else:
                    lr_groups.append(grp(self._pred, self.P, 0, self.pred_grid, row_div=s,
                                         table=self._axis_pair(self.pred_grid, (hl, wl))))
            if self.static_in_lr:
                lr_groups.append(grp(self._stat, self.S, 2, (H, W), origin_from_crop=1,
                                     table=self._axis_pair((self.psy, self.psx), (oy, ox))))
        oy, ox = self.lr_out
        self._passes.append(dict(groups=lr_groups, crop=True, out='lr', oh=oy, ow=ox, T=T))
        self._passes.append(dict(groups=[grp(self._hr, self.C, 0, (H, W), raw=1)], crop=True, out='hr', oh=self.psy, ow=self.psx, T=T))
        if self.S:
            self._passes.append(dict(groups=[grp(self._stat, self.S, 2, (H, W), raw=1)], crop=True, out='st', oh=self.psy,
                                     ow=self.psx, T=1))

    def _run_general(self, idx, cy, cx, lr, hr, st):
        import ctypes
        from . import _lib

        class TapAxis(ctypes.Structure):
            _fields_ = [('idx', ctypes.c_void_p), ('wt', ctypes.c_void_p), ('k', ctypes.c_int)]

        class Group(ctypes.Structure):
            _fields_ = [('src', ctypes.c_void_p), ('channels', ctypes.c_int), ('frames', ctypes.c_int), ('src_h', ctypes.c_int),
                        ('src_w', ctypes.c_int), ('raw', ctypes.c_int), ('origin_from_crop', ctypes.c_int), ('row_div', ctypes.c_int),
                        ('taps', TapAxis * 2)]
        ip = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data
        outs = dict(scratch=self._scratch_g, lr=lr, hr=hr, st=st)
        for ps in self._passes:
            arr = (Group * len(ps['groups']))()
            for i, g in enumerate(ps['groups']):
                arr[i].src, arr[i].channels, arr[i].frames = g['src'].ptr, g['channels'], g['frames']
                arr[i].src_h, arr[i].src_w = g['grid']
                arr[i].raw, arr[i].origin_from_crop, arr[i].row_div = g['raw'], g['origin_from_crop'], g['row_div']
                if g['table'] is not None:
                    for k in range(2):
                        arr[i].taps[k].idx, arr[i].taps[k].wt, arr[i].taps[k].k = g['table'][k].idx, g['table'][k].wt, g['table'][k].k
            _lib.check(_lib.lib().dl4ds_batch_gather(ctypes.addressof(arr), len(ps['groups']), ip(idx),
                                                     ip(cy) if ps['crop'] else None, ip(cx) if ps['crop'] else None,
                                                     outs[ps['out']].ptr, ps['oh'], ps['ow'], ps['T'], self.batch_size))

    def __len__(self):
        return len(self.indices) // self.batch_size

    def _draw(self, index):
        return self._draw_for(self.indices[index * self.batch_size:(index + 1) * self.batch_size])

    def _draw_for(self, sample_indices):
        """Crop corners (HR pixels) for the given samples, with the RNG calls of create_pair_hr_lr / crop_array: 'pin' and
        post-upsampling without predictors crop the HR field at any pixel (dataloader.py:107-112,201-205); with predictors
        or a caller-supplied LR array the corner is drawn on the LR grid and scaled (dataloader.py:166-174,193-200)."""
        idx = np.asarray(sample_indices, np.int32)
        cy = np.zeros(len(idx), np.int32)
        cx = np.zeros(len(idx), np.int32)
        if self.patch_size is not None:
            for b in range(len(idx)):
                if self.pin or (self.P == 0 and self._lr_src is None):
                    cy[b], cx[b] = random_corner(self.H, self.W, self.patch_size, self.rng)
                else:
                    ps_lr = self.patch_size // self.scale
                    y, x = random_corner(self.hl, self.wl, ps_lr, self.rng)
                    cy[b], cx[b] = y * self.scale, x * self.scale
        return idx, cy, cx

    def __getitem__(self, index):
        return self.prepare(self.indices[index * self.batch_size:(index + 1) * self.batch_size])

    def prepare(self, sample_indices):
        """One batch for an explicit list of `batch_size` sample indices (the CGAN loop shards its own indices)."""
        from . import _lib
        idx, cy, cx = self._draw_for(sample_indices)
        if len(idx) != self.batch_size:
            raise IndexError('incomplete batch')
        lr, hr, st = self._bufs[self._turn]
        self._turn ^= 1
        ip = lambda a: np.ascontiguousarray(a, np.int32).ctypes.data
        if self.general:
            self._run_general(idx, cy, cx, lr, hr, st)
            return ([lr, st], [hr]) if st is not None else ([lr], [hr])
        if self.taps:
            import ctypes
            ap = lambda t: None if t is None else ctypes.addressof(t)
            _lib.check(_lib.lib().dl4ds_batch_prepare_taps(
                self._hr.ptr, None if self._pred is None else self._pred.ptr, None if self._stat is None else self._stat.ptr,
                ip(idx), ip(cy), ip(cx), lr.ptr, hr.ptr, None if st is None else st.ptr,
                None if self._scratch is None else self._scratch.ptr, self.H, self.W, self.C, self.P, self.S, self.T,
                self.batch_size, self.scale, self.psy, self.psx, int(self.pin), int(self.static_in_lr), ap(self._dn_patch),
                ap(self._dn_field), ap(self._up_field)))
            return ([lr, st], [hr]) if st is not None else ([lr], [hr])
        _lib.check(_lib.lib().dl4ds_batch_prepare(
            self._hr.ptr, None if self._pred is None else self._pred.ptr, None if self._stat is None else self._stat.ptr,
            ip(idx), ip(cy), ip(cx), lr.ptr, hr.ptr, None if st is None else st.ptr, self.H, self.W, self.C, self.P, self.S,
            self.T, self.batch_size, self.scale, self.psy, self.psx, int(self.pin), int(self.static_in_lr)))
        if st is not None:
            return [lr, st], [hr]
        return [lr], [hr]
