"""numpy-in / numpy-out wrappers over the single-op C-ABI entry points (dl4ds_op_*).

Each call uploads its operands to HBM, runs the gfx950 kernel and downloads the result -- meant for
unit tests and experiments, not for speed (the train step runs through the graph runtime instead).
Layouts follow Keras: activations NHWC, conv kernels HWIO, transposed-conv kernels HWOI.
"""
import ctypes
import numpy as np
from . import _lib
from .device import DeviceArray

LOSS_KINDS = {'mae': 0, 'mse': 1, 'dssim': 2, 'dssim_mae': 3, 'dssim_mse': 4, 'dssim_mae_mse': 5,
              'msdssim': 6, 'msdssim_mae': 7, 'msdssim_mae_mse': 8}


def _d(a):
    return None if a is None else DeviceArray.from_numpy(np.asarray(a, np.float32))


def _p(d):
    return None if d is None else d.ptr


def conv2d(x, w, b=None, add=None, relu=False, d2s=0):
    """y = [relu](conv_same(x,w)+b+add) [-> depth_to_space(d2s)]"""
    n, h, wd, cin = x.shape
    ks, _, ci, cout = w.shape
    assert ci == cin
    dx, dw, db, da = _d(x), _d(w), _d(b), _d(add)
    if d2s > 1:
        y = DeviceArray.zeros((n, h * d2s, wd * d2s, cout // (d2s * d2s)))
    else:
        y = DeviceArray.zeros((n, h, wd, cout))
    _lib.check(_lib.lib().dl4ds_op_conv2d_fwd(dx.ptr, dw.ptr, _p(db), _p(da), y.ptr, n, h, wd, cin, cout, ks,
                                              int(relu), int(d2s)))
    return y.numpy()


def conv2d_epilogue(x, w, b=None, add=None, mask=None, relu=False, accumulate_into=None):
    """y = [y_old +] where(mask > 0, [relu](conv_same(x,w) + b + add), 0): every operand of the fused epilogue."""
    n, h, wd, cin = x.shape
    ks, _, ci, cout = w.shape
    assert ci == cin
    dx, dw, db, da, dm = _d(x), _d(w), _d(b), _d(add), _d(mask)
    y = DeviceArray.zeros((n, h, wd, cout)) if accumulate_into is None else _d(accumulate_into)
    _lib.check(_lib.lib().dl4ds_op_conv2d_epilogue(dx.ptr, dw.ptr, _p(db), _p(da), _p(dm), y.ptr, n, h, wd, cin, cout, ks,
                                                   int(relu), int(accumulate_into is not None)))
    return y.numpy()


def conv2d_dgrad(dz, w, d2s=0, accumulate_into=None):
    """dx = dgrad(dz, w).  dz: gradient of the conv output (in d2s layout if d2s>1)."""
    ks, _, cin, cout = w.shape
    if d2s > 1:
        n, h, wd = dz.shape[0], dz.shape[1] // d2s, dz.shape[2] // d2s
    else:
        n, h, wd = dz.shape[:3]
    ddz, dw = _d(dz), _d(w)
    dx = DeviceArray.zeros((n, h, wd, cin)) if accumulate_into is None else _d(accumulate_into)
    _lib.check(_lib.lib().dl4ds_op_conv2d_dgrad(ddz.ptr, dw.ptr, dx.ptr, n, h, wd, cin, cout, ks, int(d2s),
                                                int(accumulate_into is not None)))
    return dx.numpy()


def conv2d_wgrad(x, dz, ks, d2s=0, accumulate_into=None):
    n, h, wd, cin = x.shape
    cout = dz.shape[-1] * (d2s * d2s if d2s > 1 else 1)
    dx, ddz = _d(x), _d(dz)
    dw = DeviceArray.zeros((ks, ks, cin, cout)) if accumulate_into is None else _d(accumulate_into)
    _lib.check(_lib.lib().dl4ds_op_conv2d_wgrad(dx.ptr, ddz.ptr, dw.ptr, n, h, wd, cin, cout, ks, int(d2s),
                                                int(accumulate_into is not None)))
    return dw.numpy()


def bias_act_bwd(dy, y=None, want_db=True):
    n, h, w, c = dy.shape
    ddy, dyy = _d(dy), _d(y)
    db = DeviceArray.zeros((c,)) if want_db else None
    _lib.check(_lib.lib().dl4ds_op_bias_act_bwd(ddy.ptr, _p(dyy), _p(db), n, h, w, c))
    return ddy.numpy(), (db.numpy() if want_db else None)


def conv2d_transpose(x, w, stride, relu=False):
    n, h, wd, cin = x.shape
    ks, _, cout, ci = w.shape
    assert ci == cin
    dx, dw = _d(x), _d(w)
    y = DeviceArray.zeros((n, h * stride, wd * stride, cout))
    _lib.check(_lib.lib().dl4ds_op_conv2d_transpose_fwd(dx.ptr, dw.ptr, y.ptr, n, h, wd, cin, cout, ks, stride, int(relu)))
    return y.numpy()


def conv2d_transpose_dgrad(dz, w, stride):
    ks, _, cout, cin = w.shape
    n, h, wd = dz.shape[0], dz.shape[1] // stride, dz.shape[2] // stride
    ddz, dw = _d(dz), _d(w)
    dx = DeviceArray.zeros((n, h, wd, cin))
    _lib.check(_lib.lib().dl4ds_op_conv2d_transpose_dgrad(ddz.ptr, dw.ptr, dx.ptr, n, h, wd, cin, cout, ks, stride, 0))
    return dx.numpy()


def conv2d_transpose_wgrad(x, dz, ks, stride):
    n, h, wd, cin = x.shape
    cout = dz.shape[-1]
    dx, ddz = _d(x), _d(dz)
    dw = DeviceArray.zeros((ks, ks, cout, cin))
    _lib.check(_lib.lib().dl4ds_op_conv2d_transpose_wgrad(dx.ptr, ddz.ptr, dw.ptr, n, h, wd, cin, cout, ks, stride, 0))
    return dw.numpy()


def depth_to_space(x, r):
    n, h, w, c = x.shape
    dx = _d(x)
    y = DeviceArray.zeros((n, h * r, w * r, c // (r * r)))
    _lib.check(_lib.lib().dl4ds_op_depth_to_space(dx.ptr, y.ptr, n, h, w, c, r))
    return y.numpy()


def space_to_depth(y, r):
    n, hh, ww, cp = y.shape
    h, w, c = hh // r, ww // r, cp * r * r
    dy = _d(y)
    x = DeviceArray.zeros((n, h, w, c))
    _lib.check(_lib.lib().dl4ds_op_space_to_depth(dy.ptr, x.ptr, n, h, w, c, r))
    return x.numpy()


def maxpool2(x):
    n, h, w, c = x.shape
    dx = _d(x)
    y = DeviceArray.zeros((n, h // 2, w // 2, c))
    _lib.check(_lib.lib().dl4ds_op_maxpool2_fwd(dx.ptr, y.ptr, n, h, w, c))
    return y.numpy()


def maxpool2_bwd(x, y, dy):
    n, h, w, c = x.shape
    dx_, dy_, ddy = _d(x), _d(y), _d(dy)
    dx = DeviceArray.zeros(x.shape)
    _lib.check(_lib.lib().dl4ds_op_maxpool2_bwd(dx_.ptr, dy_.ptr, ddy.ptr, dx.ptr, n, h, w, c))
    return dx.numpy()


def dwconv(x, k, bias=None, dy=None, accumulate_into=None):
    """DepthwiseConv2D(7, 'same')(x); with ``dy`` also (dx, dk, db).  k: (7,7,C,1) or (7,7,C)."""
    n, h, w, c = x.shape
    ks = k.shape[0]
    dx_, dk_ = _d(x), _d(np.ascontiguousarray(k, np.float32).reshape(ks, ks, c))
    db_ = None if bias is None else _d(bias)
    y = DeviceArray.zeros(x.shape)
    _lib.check(_lib.lib().dl4ds_op_dwconv_fwd(dx_.ptr, dk_.ptr, None if db_ is None else db_.ptr, y.ptr, n, h, w, c, ks))
    if dy is None:
        return y.numpy()
    ddy = _d(dy)
    gx = DeviceArray.zeros(x.shape) if accumulate_into is None else _d(accumulate_into)
    gk, gb = DeviceArray.zeros((ks, ks, c)), DeviceArray.zeros((c,))
    _lib.check(_lib.lib().dl4ds_op_dwconv_bwd(dx_.ptr, dk_.ptr, ddy.ptr, gx.ptr, gk.ptr, gb.ptr, n, h, w, c, ks,
                                              int(accumulate_into is not None)))
    return y.numpy(), gx.numpy(), gk.numpy().reshape(k.shape), gb.numpy()


def layernorm(x, gamma, beta, eps=1e-3, relu=False, dy=None):
    """LayerNormalization(axis=-1)(x) [+ ReLU]; with ``dy`` also (dx, dgamma, dbeta)."""
    c = x.shape[-1]
    npix = x.size // c
    dx_, dg_, db_ = _d(x), _d(gamma), _d(beta)
    y = DeviceArray.zeros(x.shape)
    _lib.check(_lib.lib().dl4ds_op_layernorm_fwd(dx_.ptr, dg_.ptr, db_.ptr, y.ptr, npix, c, float(eps), int(relu)))
    if dy is None:
        return y.numpy()
    ddy = _d(dy)
    dx, dgam, dbet = DeviceArray.zeros(x.shape), DeviceArray.zeros((c,)), DeviceArray.zeros((c,))
    _lib.check(_lib.lib().dl4ds_op_layernorm_bwd(dx_.ptr, y.ptr, ddy.ptr, dg_.ptr, dx.ptr, dgam.ptr, dbet.ptr, npix, c,
                                                 float(eps), int(relu), 0))
    return y.numpy(), dx.numpy(), dgam.numpy(), dbet.numpy()


def batchnorm(x, gamma, beta, moving_mean, moving_var, eps=1e-3, momentum=0.99, training=True, relu=False, dy=None):
    """BatchNormalization(axis=-1)(x, training) [+ ReLU] -> (y, new moving_mean, new moving_var[, dx, dgamma, dbeta])."""
    c = x.shape[-1]
    npix = x.size // c
    dx_, dg_, db_, dmm, dmv = _d(x), _d(gamma), _d(beta), _d(moving_mean), _d(moving_var)
    y, saved = DeviceArray.zeros(x.shape), DeviceArray.zeros((2 * c,))
    _lib.check(_lib.lib().dl4ds_op_batchnorm_fwd(dx_.ptr, dg_.ptr, db_.ptr, dmm.ptr, dmv.ptr, y.ptr, saved.ptr, npix, c,
                                                 float(eps), float(momentum), int(training), int(relu)))
    if dy is None:
        return y.numpy(), dmm.numpy(), dmv.numpy()
    ddy = _d(dy)
    dx, dgam, dbet = DeviceArray.zeros(x.shape), DeviceArray.zeros((c,)), DeviceArray.zeros((c,))
    _lib.check(_lib.lib().dl4ds_op_batchnorm_bwd(dx_.ptr, y.ptr, ddy.ptr, dg_.ptr, saved.ptr, dx.ptr, dgam.ptr, dbet.ptr,
                                                 npix, c, int(relu), 0))
    return y.numpy(), dmm.numpy(), dmv.numpy(), dx.numpy(), dgam.numpy(), dbet.numpy()


def resize_bilinear(x, ho, wo):
    n, h, w, c = x.shape
    dx = _d(x)
    y = DeviceArray.zeros((n, ho, wo, c))
    _lib.check(_lib.lib().dl4ds_op_resize_bilinear_fwd(dx.ptr, y.ptr, n, h, w, c, ho, wo))
    return y.numpy()


def resize_bilinear_bwd(dy, h, w):
    n, ho, wo, c = dy.shape
    ddy = _d(dy)
    dx = DeviceArray.zeros((n, h, w, c))
    _lib.check(_lib.lib().dl4ds_op_resize_bilinear_bwd(ddy.ptr, dx.ptr, n, h, w, c, ho, wo))
    return dx.numpy()


def localconv(x, w, b):
    n, h, wd, c = x.shape
    f = w.shape[-1]
    dx, dw, db = _d(x), _d(w), _d(b)
    y = DeviceArray.zeros((n, h, wd, f))
    _lib.check(_lib.lib().dl4ds_op_localconv_fwd(dx.ptr, dw.ptr, _p(db), y.ptr, n, h, wd, c, f))
    return y.numpy()


def localconv_bwd(x, w, dy):
    n, h, wd, c = x.shape
    f = w.shape[-1]
    dx_, dw_, ddy = _d(x), _d(w), _d(dy)
    dx = DeviceArray.zeros(x.shape)
    dw = DeviceArray.zeros(w.shape)
    db = DeviceArray.zeros((h, wd, f))
    _lib.check(_lib.lib().dl4ds_op_localconv_bwd(dx_.ptr, dw_.ptr, ddy.ptr, dx.ptr, dw.ptr, db.ptr, n, h, wd, c, f))
    return dx.numpy(), dw.numpy(), db.numpy()


def _att_shape(x):
    if x.ndim == 4:
        b, h, w, c = x.shape
        return b, h * w, 1, c
    b, t, h, w, c = x.shape
    return b, t * h, w, c


def channel_attention(x, w1, b1, w2, b2, return_saved=False):
    g, r, p, c = _att_shape(x)
    cr = w1.shape[-1]
    dx = _d(x)
    ws = [_d(np.asarray(v).reshape(-1)) for v in (w1, b1, w2, b2)]
    y = DeviceArray.zeros(x.shape)
    saved = DeviceArray.zeros((g * p * (2 * c + cr),))
    _lib.check(_lib.lib().dl4ds_op_chatt_fwd(dx.ptr, y.ptr, g, r, p, c, cr, ws[0].ptr, ws[1].ptr, ws[2].ptr,
                                             ws[3].ptr, saved.ptr))
    if return_saved:
        return y.numpy(), saved
    return y.numpy()


def channel_attention_bwd(x, dy, w1, b1, w2, b2):
    g, r, p, c = _att_shape(x)
    cr = w1.shape[-1]
    _, saved = channel_attention(x, w1, b1, w2, b2, return_saved=True)
    dx_, ddy = _d(x), _d(dy)
    dw1_, dw2_ = _d(np.asarray(w1).reshape(-1)), _d(np.asarray(w2).reshape(-1))
    dx = DeviceArray.zeros(x.shape)
    g1, gb1, g2, gb2 = (DeviceArray.zeros((c * cr,)), DeviceArray.zeros((cr,)), DeviceArray.zeros((cr * c,)),
                        DeviceArray.zeros((c,)))
    _lib.check(_lib.lib().dl4ds_op_chatt_bwd(dx_.ptr, ddy.ptr, dx.ptr, g, r, p, c, cr, dw1_.ptr, dw2_.ptr, saved.ptr,
                                             g1.ptr, gb1.ptr, g2.ptr, gb2.ptr))
    return dx.numpy(), g1.numpy().reshape(np.shape(w1)), gb1.numpy(), g2.numpy().reshape(np.shape(w2)), gb2.numpy()


def loss(kind, y_true, y_pred, want_grad=True):
    n, h, w, c = y_true.shape
    dt, dp = _d(y_true), _d(y_pred)
    g = DeviceArray.zeros(y_pred.shape) if want_grad else None
    lv = DeviceArray.zeros((8,))
    _lib.check(_lib.lib().dl4ds_op_loss(LOSS_KINDS[kind], dt.ptr, dp.ptr, _p(g), n, h, w, c, lv.ptr))
    return float(lv.numpy()[0]), (g.numpy() if want_grad else None)


def bce(p, label):
    dp = _d(np.asarray(p, np.float32).reshape(-1))
    g = DeviceArray.zeros((dp.shape[0],))
    lv = DeviceArray.zeros((8,))
    _lib.check(_lib.lib().dl4ds_op_bce(dp.ptr, float(label), dp.shape[0], lv.ptr, g.ptr))
    return float(lv.numpy()[0]), g.numpy().reshape(np.shape(p))


def adam(w, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0):
    dw, dg, dm, dv = (_d(np.asarray(a).reshape(-1)) for a in (w, g, m, v))
    _lib.check(_lib.lib().dl4ds_op_adam(dw.ptr, dg.ptr, dm.ptr, dv.ptr, dw.shape[0], int(t), float(lr), float(beta1),
                                        float(beta2), float(eps), float(grad_scale)))
    sh = np.shape(w)
    return dw.numpy().reshape(sh), dm.numpy().reshape(sh), dv.numpy().reshape(sh)
