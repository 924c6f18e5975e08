"""Plain-numpy restatement of the TF/Keras primitives reached from dl4ds's
hot path (forward only).  TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

Layouts (Keras defaults): activations NHWC, Conv2D kernel HWIO
``w[ky,kx,cin,cout]``, Conv2DTranspose kernel HWOI ``w[ky,kx,cout,cin]``,
Dense ``w[in,out]``.  dtype follows the inputs (float32 or float64).

Every function names the reference call site that depends on it
(paths relative to /root/reference).
"""
import numpy as np

name = 'numpy'


# ----------------------------------------------------------------------------
# helpers
def asarray(x, dtype=None):
    return np.asarray(x, dtype=dtype)


def to_numpy(x):
    return np.asarray(x)


def same_pad(size, k, s):
    """TF 'SAME' padding: out=ceil(in/s); total=max((out-1)*s+k-in,0);
    before=total//2 (the extra cell goes bottom/right)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


# ----------------------------------------------------------------------------
# convolutions
def conv2d(x, w, b=None, stride=1, padding='same'):
    """tf.keras.layers.Conv2D (cross-correlation, zero padding).
    Used by: dl4ds/models/blocks.py:49-61,208,249-259,299,414-416,479,582-583;
    dl4ds/models/sp_postups.py:134,156; dl4ds/models/discriminator.py:35-65.
    Rank-5 inputs fold the leading dims into the batch (Keras behaviour)."""
    lead = None
    if x.ndim == 5:
        lead = x.shape[:2]
        x = x.reshape((-1,) + x.shape[2:])
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    assert ci == c, (ci, c)
    if padding == 'same':
        ho, pt, pb = same_pad(h, kh, stride)
        wo, pl, pr = same_pad(wd, kw, stride)
    else:
        ho = (h - kh) // stride + 1
        wo = (wd - kw) // stride + 1
        pt = pb = pl = pr = 0
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((n, ho, wo, co), dtype=np.result_type(x.dtype, w.dtype))
    for ky in range(kh):
        for kx in range(kw):
            patch = xp[:, ky:ky + (ho - 1) * stride + 1:stride,
                       kx:kx + (wo - 1) * stride + 1:stride, :]
            y += np.tensordot(patch, w[ky, kx], axes=([3], [0]))
    if b is not None:
        y = y + b
    if lead is not None:
        y = y.reshape(lead + y.shape[1:])
    return y


def conv2d_transpose(x, w, stride):
    """tf.keras.layers.Conv2DTranspose(padding='same', use_bias=False),
    dl4ds/models/blocks.py:508-516.  Kernel HWOI.  out = in*stride;
    y[n, i*s+ky-pb, j*s+kx-pb, o] += x[n,i,j,c] * w[ky,kx,o,c],
    pb = (k-s)//2 -- the exact adjoint of the SAME strided Conv2D that maps the
    out-grid back to the in-grid."""
    n, h, wd, c = x.shape
    kh, kw, co, ci = w.shape
    assert ci == c
    s = stride
    ho, wo = h * s, wd * s
    pbh = max(kh - s, 0) // 2
    pbw = max(kw - s, 0) // 2
    full = np.zeros((n, (h - 1) * s + kh, (wd - 1) * s + kw, co),
                    dtype=np.result_type(x.dtype, w.dtype))
    for ky in range(kh):
        for kx in range(kw):
            contrib = np.tensordot(x, w[ky, kx], axes=([3], [1]))  # n,h,w,co
            full[:, ky:ky + (h - 1) * s + 1:s, kx:kx + (wd - 1) * s + 1:s, :] += contrib
    # crop; if k < s the full canvas is smaller than out -> pad with zeros
    y = np.zeros((n, ho, wo, co), dtype=full.dtype)
    src = full[:, pbh:pbh + ho, pbw:pbw + wo, :]
    y[:, :src.shape[1], :src.shape[2], :] = src
    return y


def depth_to_space(x, r):
    """tf.nn.depth_to_space NHWC ("DCR"), dl4ds/models/blocks.py:427:
    y[n, h*r+i, w*r+j, c] = x[n, h, w, (i*r+j)*C' + c]."""
    n, h, w, c = x.shape
    cp = c // (r * r)
    y = x.reshape(n, h, w, r, r, cp).transpose(0, 1, 3, 2, 4, 5)
    return y.reshape(n, h * r, w * r, cp)


def resize_nearest(x, ho, wo):
    """tf.image.resize(method='nearest') (half_pixel_centers=True): src = min(floor((dst + 0.5) * in / out), in - 1)."""
    h, w = x.shape[1], x.shape[2]
    iy = np.minimum(np.floor((np.arange(ho) + 0.5) * (h / ho)).astype(int), h - 1)
    ix = np.minimum(np.floor((np.arange(wo) + 0.5) * (w / wo)).astype(int), w - 1)
    return x[:, iy][:, :, ix]


def resize_bilinear(x, ho, wo):
    """tf.keras.layers.Resizing(..., 'bilinear') = tf.image.resize, half-pixel
    centres, no antialias.  dl4ds/models/blocks.py:489; discriminator.py:62-63.
    src = (dst+0.5)*in/out - 0.5; lower=max(floor(src),0); upper=min(ceil(src),in-1);
    lerp = src - floor(src)  (TF's compute_interpolation_weights)."""
    n, h, w, c = x.shape

    def weights(out, inn):
        scale = inn / out
        src = (np.arange(out, dtype=np.float64) + 0.5) * scale - 0.5
        f = np.floor(src)
        lo = np.maximum(f, 0).astype(np.int64)
        hi = np.minimum(np.ceil(src), inn - 1).astype(np.int64)
        return lo, hi, (src - f).astype(x.dtype)

    y0, y1, fy = weights(ho, h)
    x0, x1, fx = weights(wo, w)
    top = x[:, y0][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y0][:, :, x1] * fx[None, None, :, None]
    bot = x[:, y1][:, :, x0] * (1 - fx)[None, None, :, None] + x[:, y1][:, :, x1] * fx[None, None, :, None]
    return top * (1 - fy)[None, :, None, None] + bot * fy[None, :, None, None]


def bicubic_axis_matrix(inn, out):
    """Dense (out, inn) interpolation matrix of tf.image.resize(method='bicubic') along one axis -- the ResizeBicubic op
    with half_pixel_centers=True (TensorFlow core/kernels/image/resize_bicubic_op.cc; TF is absent from this image, the
    op's published algorithm is restated): float32 arithmetic throughout; scale = in / out; src = (o + 0.5) * scale - 0.5;
    i0 = floor(src); Keys cubic with A = -0.5 read from a table of 1024 steps at offset = rint((src - i0) * 1024):
    w(i0) = lut0[off], w(i0-1) = lut1[off], w(i0+1) = lut0[1024-off], w(i0+2) = lut1[1024-off]; a tap whose index has to
    be clamped into the image gets weight 0 and the rest is renormalised to sum 1."""
    f32 = np.float32
    A = f32(-0.5)
    t = np.arange(1025, dtype=np.float32) / f32(1024)
    lut0 = ((A + f32(2)) * t - (A + f32(3))) * t * t + f32(1)
    t1 = t + f32(1)
    lut1 = ((A * t1 - f32(5) * A) * t1 + f32(8) * A) * t1 - f32(4) * A
    scale = f32(inn) / f32(out)
    M = np.zeros((out, inn), np.float64)
    for o in range(out):
        src = (f32(o) + f32(0.5)) * scale - f32(0.5)
        i0 = int(np.floor(src))
        off = int(np.rint((src - f32(i0)) * f32(1024)))
        taps = [(i0 - 1, lut1[off]), (i0, lut0[off]), (i0 + 1, lut0[1024 - off]), (i0 + 2, lut1[1024 - off])]
        ws = [f32(w) if 0 <= i < inn else f32(0) for i, w in taps]
        tot = f32(0)
        for w in ws:
            tot = f32(tot + w)
        if abs(tot) >= 1000 * np.finfo(np.float32).tiny:
            ws = [f32(w * (f32(1) / tot)) for w in ws]
        for (i, _), w in zip(taps, ws):
            M[o, min(max(i, 0), inn - 1)] += float(w)
    return M


def resize_bicubic(x, ho, wo):
    """tf.keras.layers.Resizing(..., 'bicubic') (dl4ds/models/blocks.py:473-489): separable, see bicubic_axis_matrix."""
    My = bicubic_axis_matrix(x.shape[1], ho)
    Mx = bicubic_axis_matrix(x.shape[2], wo)
    return np.einsum('oh,nhwc,pw->nopc', My, x, Mx)


def scale_translate_axis_matrix(inn, out, method):
    """Dense (out, inn) matrix of tf.image.resize(method in {'lanczos3', 'lanczos5', 'gaussian', 'mitchellcubic'},
    antialias=False) along one axis: the ScaleAndTranslate op with scale = out / in, translation 0 and kernel scale 1
    (TensorFlow core/kernels/image/scale_and_translate_op.cc ComputeSpans + sampling_kernels.h; TF is absent from this image,
    the op's published algorithm is restated, float32 like the op): sample = (o + 0.5) / scale; source pixels
    ceil(sample - R - 0.5) .. floor(sample + R - 0.5), clamped into the image; weight kernel(|i + 0.5 - sample|), the span
    normalised to sum 1.  Kernels: Lanczos (R = 3 / 5): 0 beyond R, 1 within 1e-3, else R sin(pi x) sin(pi x / R) / (pi x)^2;
    Gaussian: R = 1.5, sigma = R / 3; Mitchell-Netravali cubic (B = C = 1/3), R = 2."""
    f32 = np.float32
    R = {'lanczos3': 3.0, 'lanczos5': 5.0, 'gaussian': 1.5, 'mitchellcubic': 2.0}[method]

    def kernel(x):
        x = f32(abs(x))
        if method.startswith('lanczos'):
            pi = f32(3.14159265359)
            if x > R:
                return f32(0)
            if x <= f32(1e-3):
                return f32(1)
            return f32(f32(R) * np.sin(pi * x) * np.sin(pi * x / f32(R)) / (pi * pi * x * x))
        if method == 'gaussian':
            sigma = f32(R) / f32(3)
            return f32(0) if x >= R else f32(np.exp(-x * x / (f32(2) * sigma * sigma)))
        if x >= 2:
            return f32(0)
        if x >= 1:
            return f32(((f32(-7 / 18) * x + f32(2)) * x - f32(10 / 3)) * x + f32(16 / 9))
        return f32(((f32(7 / 6) * x - f32(2)) * x) * x + f32(8 / 9))
    inv_scale = f32(1) / (f32(out) / f32(inn))
    M = np.zeros((out, inn), np.float64)
    for o in range(out):
        sample = (f32(o) + f32(0.5)) * inv_scale
        if sample < 0 or sample > inn:
            continue
        s0 = int(np.ceil(sample - f32(R) - f32(0.5)))
        s1 = int(np.floor(sample + f32(R) - f32(0.5)))
        s0 = min(max(s0, 0), inn - 1)
        s1 = min(max(s1, 0), inn - 1) + 1
        ws = [kernel(f32(i) + f32(0.5) - sample) for i in range(s0, s1)]
        tot = f32(0)
        for w in ws:
            tot = f32(tot + w)
        if abs(tot) >= 1000 * np.finfo(np.float32).tiny:
            ws = [f32(w * (f32(1) / tot)) for w in ws]
        for i, w in zip(range(s0, s1), ws):
            M[o, i] += float(w)
    return M


def resize_scale_translate(x, ho, wo, method):
    """tf.keras.layers.Resizing(..., interpolation=method) for the ScaleAndTranslate family (blocks.py:473-489)."""
    My = scale_translate_axis_matrix(x.shape[1], ho, method)
    Mx = scale_translate_axis_matrix(x.shape[2], wo, method)
    return np.einsum('oh,nhwc,pw->nopc', My, x, Mx)


def max_pool2(x):
    """MaxPooling2D((2,2)) VALID stride 2.  dl4ds/models/blocks.py:613."""
    n, h, w, c = x.shape
    ho, wo = h // 2, w // 2
    v = x[:, :ho * 2, :wo * 2, :].reshape(n, ho, 2, wo, 2, c)
    return v.max(axis=(2, 4))


def locally_connected_1x1(x, w, b):
    """LocallyConnected2D(filters, (1,1), implementation=3) with bias,
    dl4ds/models/blocks.py:322-328: y[n,h,w,f] = b[h,w,f] + sum_c x[n,h,w,c] W[h,w,c,f]."""
    return np.einsum('nhwc,hwcf->nhwf', x, w) + b


# ----------------------------------------------------------------------------
# pointwise / structural
def relu(x):
    return np.maximum(x, 0)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def tanh(x):
    return np.tanh(x)


def hard_sigmoid(x):
    """tf.keras 2.x hard_sigmoid: clip(0.2x+0.5, 0, 1) (ConvLSTM2D recurrent_activation)."""
    return np.clip(0.2 * x + 0.5, 0.0, 1.0)


def activation(x, kind):
    """Keras Activation(kind); None -> linear (dl4ds/models/blocks.py:75)."""
    if kind is None or kind == 'linear':
        return x
    if kind == 'relu':
        return relu(x)
    if kind == 'sigmoid':
        return sigmoid(x)
    if kind == 'tanh':
        return np.tanh(x)
    if kind == 'elu':
        return np.where(x > 0, x, np.exp(np.minimum(x, 0)) - 1)
    if kind == 'leaky_relu':     # keras.activations.leaky_relu / tf.nn.leaky_relu alpha=0.2
        return np.where(x > 0, x, 0.2 * x)
    if kind == 'selu':
        a, s = 1.6732632423543772, 1.0507009873554805
        return s * np.where(x > 0, x, a * (np.exp(np.minimum(x, 0)) - 1))
    if kind == 'gelu':           # exact erf form
        from scipy.special import erf
        return 0.5 * x * (1 + erf(x / np.sqrt(2.0)))
    raise ValueError(kind)


def concat(xs, axis=-1):
    return np.concatenate(xs, axis=axis)


def add(a, b):
    return a + b


def pad_bottom_right(x, ph, pw):
    """ZeroPadding2D(((0,ph),(0,pw))) -- dl4ds/models/blocks.py:639-647."""
    return np.pad(x, ((0, 0), (0, ph), (0, pw), (0, 0)))


def mean_hw(x, keepdims=True):
    """tf.reduce_mean(x, axis=[1,2]) -- dl4ds/models/blocks.py:587.  NB: on 5-D
    (B,T,H,W,C) input axes [1,2] are (T,H)."""
    return x.mean(axis=(1, 2), keepdims=keepdims)


def global_avg_pool(x):
    """GlobalAveragePooling2D/3D: mean over all but batch & channel."""
    return x.mean(axis=tuple(range(1, x.ndim - 1)))


def dense(x, w, b):
    return x @ w + b


def mul(a, b):
    return a * b


def expand_repeat_time(s, t):
    """tf.expand_dims(s,1); tf.repeat(s, t, axis=1) -- spt_postups.py:139-140."""
    return np.repeat(s[:, None], t, axis=1)


def channel_attention(x, w1, b1, w2, b2):
    """ChannelAttention2D.call -- dl4ds/models/blocks.py:585-593.
    w1: (1,1,C,C//r) b1: (C//r,) w2: (1,1,C//r,C) b2: (C,)."""
    y = mean_hw(x, keepdims=True)
    y = conv2d(y, w1, b1)
    y = relu(y)
    y = conv2d(y, w2, b2)
    y = sigmoid(y)
    return x * y


def conv_lstm2d(x, kernel, rec_kernel, bias):
    """tf.keras.layers.ConvLSTM2D(F, k, return_sequences=True, padding='same'),
    tf.keras-2 defaults: activation=tanh, recurrent_activation=hard_sigmoid,
    gate order i,f,c,o along the last kernel axis, zero initial state, no
    recurrent bias.  dl4ds/models/blocks.py:350-355.
    x: (B,T,H,W,Cin); kernel (k,k,Cin,4F); rec_kernel (k,k,F,4F); bias (4F,)."""
    bsz, t, h, w, _ = x.shape
    f = rec_kernel.shape[2]
    hs = np.zeros((bsz, h, w, f), dtype=x.dtype)
    cs = np.zeros_like(hs)
    outs = []
    for ti in range(t):
        z = conv2d(x[:, ti], kernel, bias) + conv2d(hs, rec_kernel, None)
        zi, zf, zc, zo = z[..., :f], z[..., f:2 * f], z[..., 2 * f:3 * f], z[..., 3 * f:]
        i = hard_sigmoid(zi)
        fg = hard_sigmoid(zf)
        cs = fg * cs + i * np.tanh(zc)
        o = hard_sigmoid(zo)
        hs = o * np.tanh(cs)
        outs.append(hs)
    return np.stack(outs, axis=1)


def dropout_apply(x, mask, rate):
    """Inverted dropout with an injected keep-mask (1=keep): x*mask/(1-rate).
    discriminator.py:77 (always active under training=True)."""
    return x * mask / (1.0 - rate)


def dropout_noise(x, noise, rate, variant):
    """Dropout family of get_dropout_layer (blocks.py:679-706) with the noise injected:
    'vanilla' / 'spatial' -- keep mask (1 = keep), x * mask / (1 - rate); the spatial mask has one entry per
    (leading index, channel) and is broadcast over the axes in between (SpatialDropout2D: (N, C) over H, W;
    SpatialDropout3D: (B, C) over T, H, W);  'gaussian' -- x * noise with noise ~ N(1, rate / (1 - rate))."""
    noise = asarray(noise, x.dtype)
    if variant == 'gaussian':
        return x * noise.reshape(x.shape)
    if variant == 'spatial':
        shp = (x.shape[0],) + (1,) * (len(x.shape) - 2) + (x.shape[-1],)
        return x * noise.reshape(shp) / (1.0 - rate)
    return x * noise.reshape(x.shape) / (1.0 - rate)


def depthwise_conv2d(x, k, b=None):
    """tf.keras.layers.DepthwiseConv2D(padding='same', depth_multiplier=1): k (K,K,C,1), blocks.py:143-144."""
    kh, kw, c, _ = k.shape
    ph, pw = kh // 2, kw // 2
    xp = np.pad(x, ((0, 0), (ph, ph), (pw, pw), (0, 0)))
    h, w = x.shape[1], x.shape[2]
    y = np.zeros(x.shape, dtype=np.result_type(x.dtype, k.dtype))
    for ky in range(kh):
        for kx in range(kw):
            y += xp[:, ky:ky + h, kx:kx + w, :] * k[ky, kx, :, 0]
    return y if b is None else y + b


def layer_norm(x, gamma, beta, eps=1e-3):
    """tf.keras.layers.LayerNormalization(axis=-1): biased variance over the channel axis."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def channel_moments(x):
    """Batch statistics of BatchNormalization(axis=-1): mean and biased variance over all other axes."""
    axes = tuple(range(len(x.shape) - 1))
    mu = x.mean(axis=axes)
    return mu, ((x - mu) ** 2).mean(axis=axes)


def batch_norm(x, gamma, beta, mean, var, eps=1e-3):
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


# ----------------------------------------------------------------------------
# losses  (dl4ds/losses.py)
def mae(y_true, y_pred):
    """losses.py:5-11 -- mean over every element."""
    return np.abs(y_pred - y_true).mean()


def mse(y_true, y_pred):
    """losses.py:14-20."""
    return ((y_pred - y_true) ** 2).mean()


def _gauss_kernel(size=11, sigma=1.5, dtype=np.float64):
    """tf.image.ssim's _fspecial_gauss: g = softmax(-(k-mid)^2/(2 sigma^2)) outer product."""
    coords = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    g = np.exp(-(coords ** 2) / (2.0 * sigma ** 2))
    g2 = np.outer(g, g)
    g2 /= g2.sum()
    return g2.astype(dtype)


def _valid_depthwise(x, k2):
    n, h, w, c = x.shape
    k = k2.shape[0]
    ho, wo = h - k + 1, w - k + 1
    y = np.zeros((n, ho, wo, c), dtype=x.dtype)
    for ky in range(k):
        for kx in range(k):
            y += x[:, ky:ky + ho, kx:kx + wo, :] * k2[ky, kx]
    return y


def ssim(img1, img2, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image.ssim on NHWC batches -> (N,) : mean over (H-10, W-10, C) of
    luminance*contrast-structure, means by VALID depthwise gaussian filtering."""
    g = _gauss_kernel(filter_size, filter_sigma, img1.dtype)
    c1 = (k1 * max_val) ** 2
    c2 = (k2 * max_val) ** 2
    mean0 = _valid_depthwise(img1, g)
    mean1 = _valid_depthwise(img2, g)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    lum = (num0 + c1) / (den0 + c1)
    num1 = _valid_depthwise(img1 * img2, g) * 2.0
    den1 = _valid_depthwise(img1 ** 2 + img2 ** 2, g)
    c2 = c2 * 1.0   # compensation factor 1.0 in TF
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(axis=(1, 2, 3))


def dssim(y_true, y_pred):
    """losses.py:23-55."""
    maxv = max(y_true.max(), y_pred.max())
    minv = min(y_true.min(), y_pred.min())
    drange = maxv - minv
    yt = y_true - y_true.min() if y_true.min() < 0 else y_true
    yp = y_pred - y_pred.min() if y_pred.min() < 0 else y_pred
    s = ssim(yt, yp, drange)
    return ((1 - s) / 2.0).mean()


def dssim_mae(y_true, y_pred):
    """losses.py:58-64."""
    return 0.8 * dssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred)


def dssim_mse(y_true, y_pred):
    """losses.py:83-89."""
    return 0.8 * dssim(y_true, y_pred) + 0.2 * mse(y_true, y_pred)


def dssim_mae_mse(y_true, y_pred):
    """losses.py:67-80."""
    return 0.6 * dssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred) + 0.2 * mse(y_true, y_pred)


MS_POWER_FACTORS = (0.0448, 0.2856, 0.3001, 0.2363)      # losses.py:126 (four of Wang et al.'s five scales)


def _ssim_per_channel(img1, img2, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
    """tf.image's _ssim_per_channel: (mean luminance*cs, mean cs) over the VALID region, each (N, C)."""
    g = _gauss_kernel(filter_size, filter_sigma, img1.dtype)
    c1 = (k1 * max_val) ** 2
    c2 = (k2 * max_val) ** 2
    mean0 = _valid_depthwise(img1, g)
    mean1 = _valid_depthwise(img2, g)
    num0 = mean0 * mean1 * 2.0
    den0 = mean0 ** 2 + mean1 ** 2
    lum = (num0 + c1) / (den0 + c1)
    num1 = _valid_depthwise(img1 * img2, g) * 2.0
    den1 = _valid_depthwise(img1 ** 2 + img2 ** 2, g)
    cs = (num1 - num0 + c2) / (den1 - den0 + c2)
    return (lum * cs).mean(axis=(1, 2)), cs.mean(axis=(1, 2))


def _downsample2_symmetric(x):
    """tf.image.ssim_multiscale's downsampling: SYMMETRIC-pad odd sizes at the bottom/right, then 2x2 VALID average."""
    n, h, w, c = x.shape
    if (h % 2) or (w % 2):
        x = np.pad(x, ((0, 0), (0, h % 2), (0, w % 2), (0, 0)), mode='symmetric')
        n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(axis=(2, 4))


def ssim_multiscale(img1, img2, max_val, power_factors=MS_POWER_FACTORS, filter_size=11, filter_sigma=1.5, k1=0.01,
                    k2=0.03):
    """tf.image.ssim_multiscale (third-party in the reference; algorithm as published in TF's image_ops_impl.py):
    per scale k (image halved k times) cs_k = relu(mean cs); the last scale contributes relu(mean ssim) instead;
    ms_ssim[n, c] = prod_k value_k ** power_factors[k]; result = mean over channels -> (N,)."""
    imgs = [img1, img2]
    mcs = []
    for k in range(len(power_factors)):
        if k > 0:
            imgs = [_downsample2_symmetric(x) for x in imgs]
        ssim_pc, cs = _ssim_per_channel(imgs[0], imgs[1], max_val, filter_size, filter_sigma, k1, k2)
        mcs.append(np.maximum(cs, 0))
    mcs.pop()
    vals = np.stack(mcs + [np.maximum(ssim_pc, 0)], axis=-1)              # (N, C, scales)
    ms = np.prod(vals ** np.asarray(power_factors, vals.dtype), axis=-1)
    return ms.mean(axis=-1)


def msdssim(y_true, y_pred):
    """losses.py:92-130."""
    maxv = max(y_true.max(), y_pred.max())
    minv = min(y_true.min(), y_pred.min())
    drange = maxv - minv
    yt = y_true - y_true.min() if y_true.min() < 0 else y_true
    yp = y_pred - y_pred.min() if y_pred.min() < 0 else y_pred
    return ((1 - ssim_multiscale(yt, yp, drange)) / 2.0).mean()


def msdssim_mae(y_true, y_pred):
    """losses.py:133-139."""
    return 0.8 * msdssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred)


def msdssim_mae_mse(y_true, y_pred):
    """losses.py:142-149."""
    return 0.6 * msdssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred) + 0.2 * mse(y_true, y_pred)


def bce(y_true, p):
    """tf.keras.losses.BinaryCrossentropy(from_logits=False), cgan.py:546-549,567-571:
    p clipped to [eps, 1-eps], eps=1e-7; mean over all elements."""
    eps = 1e-7
    p = np.clip(p, eps, 1 - eps)
    return -(y_true * np.log(p) + (1 - y_true) * np.log(1 - p)).mean()


# ----------------------------------------------------------------------------
# optimiser
def adam_step(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """tf.keras.optimizers.Adam (supervised.py:353; cgan.py:277-278):
    m,v EMA; lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps)
    (eps OUTSIDE the bias correction, unlike torch.optim.Adam).  t is 1-based."""
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    w = w - lr_t * m / (np.sqrt(v) + eps)
    return w, m, v


def piecewise_lr(step, boundary, lr0, lr1):
    """PiecewiseConstantDecay([boundary],[lr0,lr1]) (supervised.py:340-346):
    lr0 while step <= boundary else lr1; step = optimizer.iterations (0-based)."""
    return lr0 if step <= boundary else lr1


def image_metrics(y, p):
    """dl4ds/metrics.py:166-262 restated (TEST INFRASTRUCTURE): joint range, per-pair PSNR (tf.image.psnr), SSIM
    (tf.image.ssim with max_val = range, no shift), MAE, RMSE, Pearson over the grid; per-grid-point RMSE, mean bias and
    Pearson over the pairs."""
    y = np.asarray(y, np.float64)
    p = np.asarray(p, np.float64)
    n = y.shape[0]
    drange = max(y.max(), p.max()) - min(y.min(), p.min())
    d = p - y
    mse = (d ** 2).reshape(n, -1).mean(1)
    out = dict(drange=drange, mse=mse, rmse=np.sqrt(mse), mae=np.abs(d).reshape(n, -1).mean(1),
               psnr=20 * np.log10(drange) - 10 * np.log10(mse), ssim=ssim(y, p, drange))
    yf, pf = y.reshape(n, -1), p.reshape(n, -1)
    yc, pc = yf - yf.mean(1, keepdims=True), pf - pf.mean(1, keepdims=True)
    out['pearson'] = (yc * pc).sum(1) / np.sqrt((yc ** 2).sum(1) * (pc ** 2).sum(1))
    out['rmse_map'] = np.sqrt((d ** 2).mean(0))
    out['bias_map'] = d.mean(0)
    yc, pc = y - y.mean(0), p - p.mean(0)
    with np.errstate(invalid='ignore', divide='ignore'):
        out['pearson_map'] = (yc * pc).sum(0) / np.sqrt((yc ** 2).sum(0) * (pc ** 2).sum(0))
    return out
