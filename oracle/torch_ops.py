"""torch-CPU backend of the oracle: the same primitive API as ``np_ops`` but on
torch tensors (autograd-capable, float32 or float64), written independently of
the numpy backend (library conv kernels instead of tensordot loops).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  It provides
(a) gradients for the parity tests (torch autograd over the restated graph) and
(b) the timed ``cpu_baseline`` leg of ``bench.py`` (oneDNN conv kernels on the
host cores -- the same class of kernels TF-CPU would dispatch to).

API layout = Keras layout (NHWC / HWIO / HWOI); tensors are permuted to torch's
NCHW *views* (channels_last memory) internally.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

name = 'torch'

# ---- derivative discontinuities (used by tests/parity.py only) -------------------------------------------------------
# The gradient of a ReLU network is a DISCONTINUOUS function of its inputs: a pre-activation within rounding distance
# of zero takes either branch depending on summation order, which moves every upstream gradient by that unit's whole
# contribution (observed between two fp32 evaluations of the same graph: 1e-3 ... 5e-3 of a gradient tensor's size at
# 128 x 128, where all forward values agree to 4e-7).  With KINK = +d / -d every derivative discontinuity of the graph
# is displaced by d times the magnitude of its argument -- ReLU / leaky-ReLU / SELU thresholds, the hard-sigmoid clip
# points, the sign of the MAE residual, max-pooling ties -- which changes forward values by at most that much and
# decides every near-tie one way (+d) or the other (-d).  The two gradients bracket what any evaluation whose forward
# rounding error is below d can produce; they coincide when nothing lies within d of a discontinuity.
KINK = 0.0


class kink_shift:
    def __init__(self, rel):
        self.rel = float(rel)

    def __enter__(self):
        global KINK
        self.prev, KINK = KINK, self.rel
        return self

    def __exit__(self, *exc):
        global KINK
        KINK = self.prev
        return False


def _shifted(x):
    """x minus KINK * max|x| (x itself when KINK == 0)."""
    if KINK == 0.0:
        return x
    return x - KINK * float(x.detach().abs().max())


def asarray(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(dtype)
    t = torch.from_numpy(np.ascontiguousarray(x))
    return t if dtype is None else t.to(dtype)


def to_numpy(x):
    return x.detach().cpu().numpy()


def same_pad(size, k, s):
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def conv2d(x, w, b=None, stride=1, padding='same'):
    lead = None
    if x.dim() == 5:
        lead = tuple(x.shape[:2])
        x = x.reshape((-1,) + tuple(x.shape[2:]))
    n, h, wd, c = x.shape
    kh, kw, ci, co = w.shape
    xc = _nchw(x)
    if padding == 'same':
        _, pt, pb = same_pad(h, kh, stride)
        _, pl, pr = same_pad(wd, kw, stride)
        if pt == pb and pl == pr:
            y = F.conv2d(xc, w.permute(3, 2, 0, 1).contiguous(), b, stride=stride, padding=(pt, pl))
        else:
            xc = F.pad(xc, (pl, pr, pt, pb))
            y = F.conv2d(xc, w.permute(3, 2, 0, 1).contiguous(), b, stride=stride)
    else:
        y = F.conv2d(xc, w.permute(3, 2, 0, 1).contiguous(), b, stride=stride)
    y = _nhwc(y)
    if lead is not None:
        y = y.reshape(lead + tuple(y.shape[1:]))
    return y


def conv2d_transpose(x, w, stride):
    n, h, wd, c = x.shape
    kh, kw, co, ci = w.shape
    s = stride
    pbh = max(kh - s, 0) // 2
    pbw = max(kw - s, 0) // 2
    full = F.conv_transpose2d(_nchw(x), w.permute(3, 2, 0, 1).contiguous(), stride=s)
    y = full[:, :, pbh:pbh + h * s, pbw:pbw + wd * s]
    assert y.shape[2] == h * s and y.shape[3] == wd * s, 'k < stride not supported'
    return _nhwc(y)


def depth_to_space(x, r):
    n, h, w, c = x.shape
    cp = c // (r * r)
    y = x.reshape(n, h, w, r, r, cp).permute(0, 1, 3, 2, 4, 5)
    return y.reshape(n, h * r, w * r, cp)


def resize_nearest(x, ho, wo):
    h, w = x.shape[1], x.shape[2]
    iy = torch.clamp(torch.floor((torch.arange(ho, dtype=torch.float64) + 0.5) * (h / ho)).long(), max=h - 1)
    ix = torch.clamp(torch.floor((torch.arange(wo, dtype=torch.float64) + 0.5) * (w / wo)).long(), max=w - 1)
    return x[:, iy][:, :, ix]


def resize_bicubic(x, ho, wo):
    from . import np_ops
    My = torch.as_tensor(np_ops.bicubic_axis_matrix(x.shape[1], ho), dtype=x.dtype)
    Mx = torch.as_tensor(np_ops.bicubic_axis_matrix(x.shape[2], wo), dtype=x.dtype)
    return torch.einsum('oh,nhwc,pw->nopc', My, x, Mx)


def resize_scale_translate(x, ho, wo, method):
    """ScaleAndTranslate family ('lanczos3', 'lanczos5', 'gaussian', 'mitchellcubic'): np_ops.scale_translate_axis_matrix."""
    from . import np_ops
    My = torch.as_tensor(np_ops.scale_translate_axis_matrix(x.shape[1], ho, method), dtype=x.dtype)
    Mx = torch.as_tensor(np_ops.scale_translate_axis_matrix(x.shape[2], wo, method), dtype=x.dtype)
    return torch.einsum('oh,nhwc,pw->nopc', My, x, Mx)


def resize_bilinear(x, ho, wo):
    n, h, w, c = x.shape

    def weights(out, inn):
        scale = inn / out
        src = (torch.arange(out, dtype=torch.float64) + 0.5) * scale - 0.5
        f = torch.floor(src)
        lo = torch.clamp(f, min=0).long()
        hi = torch.clamp(torch.ceil(src), max=inn - 1).long()
        return lo, hi, (src - f).to(x.dtype)

    y0, y1, fy = weights(ho, h)
    x0, x1, fx = weights(wo, w)
    fxv = fx.view(1, 1, -1, 1)
    fyv = fy.view(1, -1, 1, 1)
    r0 = x[:, y0]
    r1 = x[:, y1]
    top = r0[:, :, x0] * (1 - fxv) + r0[:, :, x1] * fxv
    bot = r1[:, :, x0] * (1 - fxv) + r1[:, :, x1] * fxv
    return top * (1 - fyv) + bot * fyv


def max_pool2(x):
    if KINK != 0.0:
        # near-ties inside a 2 x 2 window: bias the candidates by their position, one way or the other
        h, w = x.shape[-3], x.shape[-2]
        pos = (torch.arange(h, dtype=x.dtype).view(-1, 1) % 2) * 2 + (torch.arange(w, dtype=x.dtype).view(1, -1) % 2)
        x = x + (KINK * float(x.detach().abs().max()) / 3.0) * pos.unsqueeze(-1)
    return _nhwc(F.max_pool2d(_nchw(x), 2))


def locally_connected_1x1(x, w, b):
    return torch.einsum('nhwc,hwcf->nhwf', x, w) + b


def relu(x):
    return torch.relu(_shifted(x))


def sigmoid(x):
    return torch.sigmoid(x)


def tanh(x):
    return torch.tanh(x)


def hard_sigmoid(x):
    return torch.clamp(0.2 * _shifted(x) + 0.5, 0.0, 1.0)


def activation(x, kind):
    if kind is None or kind == 'linear':
        return x
    if kind == 'relu':
        return torch.relu(_shifted(x))
    if kind == 'sigmoid':
        return torch.sigmoid(x)
    if kind == 'tanh':
        return torch.tanh(x)
    if kind == 'elu':
        return F.elu(x)
    if kind == 'leaky_relu':
        return F.leaky_relu(_shifted(x), 0.2)
    if kind == 'selu':
        return F.selu(_shifted(x))
    if kind == 'gelu':
        return F.gelu(x)
    raise ValueError(kind)


def concat(xs, axis=-1):
    return torch.cat(list(xs), dim=axis)


def add(a, b):
    return a + b


def pad_bottom_right(x, ph, pw):
    return F.pad(x, (0, 0, 0, pw, 0, ph))


def mean_hw(x, keepdims=True):
    return x.mean(dim=(1, 2), keepdim=keepdims)


def global_avg_pool(x):
    return x.mean(dim=tuple(range(1, x.dim() - 1)))


def dense(x, w, b):
    return x @ w + b


def mul(a, b):
    return a * b


def expand_repeat_time(s, t):
    return s.unsqueeze(1).expand(-1, t, -1, -1, -1)


def channel_attention(x, w1, b1, w2, b2):
    y = mean_hw(x, keepdims=True)
    c = w1.shape[2]
    y = relu(y @ w1.reshape(c, -1) + b1)
    y = torch.sigmoid(y @ w2.reshape(-1, c) + b2)
    return x * y


def conv_lstm2d(x, kernel, rec_kernel, bias):
    bsz, t, h, w, _ = x.shape
    f = rec_kernel.shape[2]
    hs = torch.zeros((bsz, h, w, f), dtype=x.dtype)
    cs = torch.zeros_like(hs)
    zx = conv2d(x, kernel, bias)            # all timesteps at once (5-D fold)
    outs = []
    for ti in range(t):
        z = zx[:, ti] + conv2d(hs, rec_kernel, None)
        zi, zf, zc, zo = z[..., :f], z[..., f:2 * f], z[..., 2 * f:3 * f], z[..., 3 * f:]
        i = hard_sigmoid(zi)
        fg = hard_sigmoid(zf)
        cs = fg * cs + i * torch.tanh(zc)
        o = hard_sigmoid(zo)
        hs = o * torch.tanh(cs)
        outs.append(hs)
    return torch.stack(outs, dim=1)


def dropout_apply(x, mask, rate):
    return x * mask / (1.0 - rate)


def dropout_noise(x, noise, rate, variant):
    noise = asarray(noise, x.dtype)
    if variant == 'gaussian':
        return x * noise.reshape(x.shape)
    if variant == 'spatial':
        shp = (x.shape[0],) + (1,) * (len(x.shape) - 2) + (x.shape[-1],)
        return x * noise.reshape(shp) / (1.0 - rate)
    return x * noise.reshape(x.shape) / (1.0 - rate)


def depthwise_conv2d(x, k, b=None):
    kh, kw, c, _ = k.shape
    w = k.permute(2, 3, 0, 1)                                   # (C, 1, K, K)
    y = F.conv2d(_nchw(x), w, None, padding=(kh // 2, kw // 2), groups=c)
    y = _nhwc(y)
    return y if b is None else y + b


def layer_norm(x, gamma, beta, eps=1e-3):
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def channel_moments(x):
    axes = tuple(range(x.dim() - 1))
    mu = x.mean(dim=axes)
    return mu, ((x - mu) ** 2).mean(dim=axes)


def batch_norm(x, gamma, beta, mean, var, eps=1e-3):
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


# ----------------------------------------------------------------------------
def mae(y_true, y_pred):
    return _shifted(y_pred - y_true).abs().mean()


def mse(y_true, y_pred):
    return ((y_pred - y_true) ** 2).mean()


def _gauss_kernel(size, sigma, dtype):
    coords = torch.arange(size, dtype=torch.float64) - (size - 1) / 2.0
    g = torch.exp(-(coords ** 2) / (2.0 * sigma ** 2))
    g2 = torch.outer(g, g)
    return (g2 / g2.sum()).to(dtype)


def _valid_depthwise(x, k2):
    c = x.shape[-1]
    wk = k2.view(1, 1, *k2.shape).expand(c, 1, -1, -1)
    return _nhwc(F.conv2d(_nchw(x), wk, groups=c))


def ssim(img1, img2, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
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
    return (lum * cs).mean(dim=(1, 2, 3))


def dssim(y_true, y_pred):
    maxv = torch.maximum(y_true.max(), y_pred.max())
    minv = torch.minimum(y_true.min(), y_pred.min())
    drange = maxv - minv
    yt = y_true - y_true.min() if y_true.min() < 0 else y_true
    yp = y_pred - y_pred.min() if y_pred.min() < 0 else y_pred
    s = ssim(yt, yp, drange)
    return ((1 - s) / 2.0).mean()


MS_POWER_FACTORS = (0.0448, 0.2856, 0.3001, 0.2363)


def _ssim_per_channel(img1, img2, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03):
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
    return (lum * cs).mean(dim=(1, 2)), cs.mean(dim=(1, 2))


def _downsample2_symmetric(x):
    n, h, w, c = x.shape
    if h % 2:
        x = torch.cat([x, x[:, -1:]], dim=1)          # SYMMETRIC pad of one row repeats the edge row
    if w % 2:
        x = torch.cat([x, x[:, :, -1:]], dim=2)
    n, h, w, c = x.shape
    return x.reshape(n, h // 2, 2, w // 2, 2, c).mean(dim=(2, 4))


def ssim_multiscale(img1, img2, max_val, power_factors=MS_POWER_FACTORS, filter_size=11, filter_sigma=1.5, k1=0.01,
                    k2=0.03):
    imgs = [img1, img2]
    mcs = []
    for k in range(len(power_factors)):
        if k > 0:
            imgs = [_downsample2_symmetric(x) for x in imgs]
        ssim_pc, cs = _ssim_per_channel(imgs[0], imgs[1], max_val, filter_size, filter_sigma, k1, k2)
        mcs.append(torch.relu(cs))
    mcs.pop()
    vals = torch.stack(mcs + [torch.relu(ssim_pc)], dim=-1)
    pf = torch.as_tensor(power_factors, dtype=vals.dtype)
    return torch.prod(vals ** pf, dim=-1).mean(dim=-1)


def msdssim(y_true, y_pred):
    maxv = torch.maximum(y_true.max(), y_pred.max())
    minv = torch.minimum(y_true.min(), y_pred.min())
    drange = maxv - minv
    yt = y_true - y_true.min() if y_true.min() < 0 else y_true
    yp = y_pred - y_pred.min() if y_pred.min() < 0 else y_pred
    return ((1 - ssim_multiscale(yt, yp, drange)) / 2.0).mean()


def msdssim_mae(y_true, y_pred):
    return 0.8 * msdssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred)


def msdssim_mae_mse(y_true, y_pred):
    return 0.6 * msdssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred) + 0.2 * mse(y_true, y_pred)


def dssim_mae(y_true, y_pred):
    return 0.8 * dssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred)


def dssim_mse(y_true, y_pred):
    return 0.8 * dssim(y_true, y_pred) + 0.2 * mse(y_true, y_pred)


def dssim_mae_mse(y_true, y_pred):
    return 0.6 * dssim(y_true, y_pred) + 0.2 * mae(y_true, y_pred) + 0.2 * mse(y_true, y_pred)


def bce(y_true, p):
    eps = 1e-7
    p = torch.clamp(p, eps, 1 - eps)
    return -(y_true * torch.log(p) + (1 - y_true) * torch.log(1 - p)).mean()


def adam_step(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    w = w - lr_t * m / (torch.sqrt(v) + eps)
    return w, m, v
