"""Restatement of the per-step training arithmetic (torch-CPU autograd backend).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.

* ``supervised_step``: Keras ``fit`` inner step as configured by
  ``SupervisedTrainer.run`` (dl4ds/training/supervised.py:353,396-406):
  forward -> loss -> grads -> Adam.
* ``cgan_step``: ``train_step`` (dl4ds/training/cgan.py:575-639) with
  ``generator_loss`` (:525-553, lambda=100) and ``discriminator_loss`` (:556-572).
"""
import math
import numpy as np
import torch

from . import torch_ops as T
from . import models as M

LOSSES = {
    'mae': T.mae, 'mse': T.mse, 'dssim': T.dssim, 'dssim_mae': T.dssim_mae,
    'dssim_mse': T.dssim_mse, 'dssim_mae_mse': T.dssim_mae_mse,
    'msdssim': T.msdssim, 'msdssim_mae': T.msdssim_mae, 'msdssim_mae_mse': T.msdssim_mae_mse,
}


class Adam:
    """tf.keras Adam over a Params dict (epsilon outside the bias correction)."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def apply(self, params, grads, lr=None):
        self.t += 1
        lr = self.lr if lr is None else lr
        with torch.no_grad():
            for k in params:
                g = grads[k]
                if g is None:
                    continue
                w, m, v = T.adam_step(params[k], g, self.m[k], self.v[k], self.t, lr,
                                      self.b1, self.b2, self.eps)
                params[k].copy_(w)
                self.m[k], self.v[k] = m, v


def _grads(loss, params):
    keys = list(params.keys())
    gs = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
    return {k: g for k, g in zip(keys, gs)}


def forward(model, cfg, P, x, s=None):
    return M.MODELS[model](T, P, x, s, **cfg)


def supervised_step(model, cfg, P, x, s, y, loss='mae', opt=None, lr=None):
    """One Keras-fit step.  P: torch Params with requires_grad.  Returns
    (loss_value, grads dict, prediction)."""
    pred = forward(model, cfg, P, x, s)
    lv = LOSSES[loss](y, pred)
    grads = _grads(lv, P)
    if opt is not None:
        opt.apply(P, grads, lr)
    return float(lv.detach()), grads, pred.detach()


def cgan_step(gen_model, gen_cfg, PG, disc_cfg, PD, lr_array, hr_array, static_array,
              dropout_masks=None, px_loss='mae', optG=None, optD=None, lam=100.0):
    """cgan.py:575-639.  dropout_masks: (mask_real, mask_fake) keep-masks for the
    discriminator's Dropout(0.4) (None -> no dropout).  Returns dict of losses/grads."""
    gen = forward(gen_model, gen_cfg, PG, lr_array, static_array)
    mr, mf = (None, None) if dropout_masks is None else dropout_masks
    bn_updates = None
    if disc_cfg.get('normalization') == 'bn':
        # two training-mode calls (cgan.py:599-600): each normalises with its own batch statistics; the second call
        # starts from the moving averages the first one left (they do not enter the training-mode output)
        c_real, c_fake = M.Ctx(training=True), M.Ctx(training=True)
        d_real = M.residual_discriminator(T, PD, lr_array, hr_array, mr, ctx=c_real, **disc_cfg)
        PD2 = M.Params()
        for k, v in PD.items():
            PD2[k] = v
        for name, (mm, mv) in c_real.bn_updates.items():
            PD2[name + '/moving_mean'], PD2[name + '/moving_variance'] = mm.detach(), mv.detach()
        d_fake = M.residual_discriminator(T, PD2, lr_array, gen, mf, ctx=c_fake, **disc_cfg)
        bn_updates = {k: (a.detach(), b.detach()) for k, (a, b) in c_fake.bn_updates.items()}
    else:
        d_real = M.residual_discriminator(T, PD, lr_array, hr_array, mr, **disc_cfg)
        d_fake = M.residual_discriminator(T, PD, lr_array, gen, mf, **disc_cfg)
    gan_loss = T.bce(torch.ones_like(d_fake), d_fake)
    px = LOSSES[px_loss](hr_array, gen)
    g_total = gan_loss + lam * px
    d_loss = T.bce(torch.ones_like(d_real), d_real) + T.bce(torch.zeros_like(d_fake), d_fake)
    gG = torch.autograd.grad(g_total, list(PG.values()), retain_graph=True, allow_unused=True)
    gD = torch.autograd.grad(d_loss, list(PD.values()), allow_unused=True)
    gG = dict(zip(PG.keys(), gG))
    gD = dict(zip(PD.keys(), gD))
    if optG is not None:
        optG.apply(PG, gG)
    if optD is not None:
        optD.apply(PD, gD)
    return dict(gen_total=float(g_total.detach()), gen_gan=float(gan_loss.detach()),
                gen_px=float(px.detach()), disc=float(d_loss.detach()),
                gradsG=gG, gradsD=gD, gen=gen.detach(),
                d_real=d_real.detach(), d_fake=d_fake.detach(), bn_updates=bn_updates)


def synthetic_batch(seed, batch, hr, scale, n_pred=0, dtype=np.float32):
    """SURVEY.md section 8d synthetic inputs: HR y in U[0,1) box-blurred 5x5;
    LR x = scale x scale block mean of y.  Returns (x_lr, y_hr) NHWC."""
    rng = np.random.default_rng(seed)
    h = w = hr
    raw = rng.random((batch, h + 4, w + 4, 1 + n_pred))
    cs = raw.cumsum(axis=1).cumsum(axis=2)
    cs = np.pad(cs, ((0, 0), (1, 0), (1, 0), (0, 0)))
    blur = (cs[:, 5:, 5:] - cs[:, :-5, 5:] - cs[:, 5:, :-5] + cs[:, :-5, :-5]) / 25.0
    y = blur[..., :1]
    lr = blur.reshape(batch, h // scale, scale, w // scale, scale, -1).mean(axis=(2, 4))
    return lr.astype(dtype), y.astype(dtype)
