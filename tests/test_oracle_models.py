"""Oracle model restatements: parameter counts vs SURVEY.md (computed from the reference
builders), numpy-vs-torch forward agreement, fp64 finite-difference gradient checks."""
import numpy as np
import pytest
import torch

from oracle import np_ops as N
from oracle import torch_ops as T
from oracle import models as M
from oracle import train as TR


def nparams(P):
    return sum(int(np.prod(v.shape)) for v in P.values())


def test_param_counts_match_survey():
    # cfg2 204 405, cfg1 121 341, cfg5 G 13 566 325, D 16 177, cfg4 480 056
    P = M.init_params('net_postupsampling', (1, 8, 8, 1), backbone_block='resnet',
                      upsampling='spc', scale=4)
    assert nparams(P) == 204405
    P = M.init_params('net_pin', (1, 8, 8, 2), backbone_block='resnet')
    assert nparams(P) == 121341
    P = M.init_params('unet_pin', (1, 128, 128, 5), (1, 128, 128, 1), n_filters=8, n_blocks=6,
                      decoder_upsampling='dc')
    assert nparams(P) == 13566325
    PD = M.Params(create=True)
    M.residual_discriminator(N, PD, np.zeros((1, 8, 8, 5), np.float32),
                             np.zeros((1, 8, 8, 1), np.float32), upsampling='pin', scale=8)
    assert nparams(PD) == 16177
    P = M.init_params('recnet_postupsampling', (1, 2, 8, 8, 1), (1, 32, 32, 1),
                      backbone_block='densenet', upsampling='rc', scale=4, time_window=2,
                      attention=True, localcon_layer=True)
    # LCB params depend on the HR grid: at 256x256 they are 393 216; here 32x32 -> 6 144
    assert nparams(P) - 32 * 32 * 6 + 256 * 256 * 6 == 480056


CASES = [
    ('net_postupsampling', (2, 6, 5, 1), None, dict(backbone_block='resnet', upsampling='spc', scale=4, n_blocks=2)),
    ('net_postupsampling', (2, 6, 5, 2), (2, 12, 10, 1), dict(backbone_block='densenet', upsampling='rc', scale=2, n_blocks=2, attention=True, localcon_layer=True)),
    ('net_postupsampling', (1, 4, 4, 1), None, dict(backbone_block='convnet', upsampling='dc', scale=2, n_blocks=1)),
    ('net_pin', (2, 8, 8, 2), None, dict(backbone_block='resnet', n_blocks=2)),
    ('unet_pin', (1, 16, 16, 3), (1, 16, 16, 1), dict(n_filters=4, n_blocks=2, decoder_upsampling='dc')),
    ('unet_pin', (1, 16, 12, 3), None, dict(n_filters=4, n_blocks=2, decoder_upsampling='spc')),
    ('recnet_postupsampling', (1, 3, 6, 6, 1), (1, 12, 12, 1), dict(backbone_block='densenet', upsampling='rc', scale=2, time_window=3, n_filters=4, n_blocks=1, attention=True, localcon_layer=True)),
    ('recnet_pin', (1, 2, 6, 6, 2), None, dict(backbone_block='resnet', time_window=2, n_filters=4, n_blocks=1)),
]


@pytest.mark.parametrize('model,xs,ss,cfg', CASES)
def test_forward_numpy_vs_torch(model, xs, ss, cfg):
    P = M.init_params(model, xs, ss, seed=3, dtype=np.float64, **cfg)
    rng = np.random.default_rng(1)
    for k in P:                          # non-zero biases so they are exercised
        if P[k].ndim == 1 or 'bias' in k:
            P[k] = rng.standard_normal(P[k].shape) * 0.1
    x = rng.standard_normal(xs)
    s = None if ss is None else rng.standard_normal(ss)
    a = M.MODELS[model](N, P, x, s, **cfg)
    PT = M.convert(P, T)
    t = M.MODELS[model](T, PT, T.asarray(x), None if s is None else T.asarray(s), **cfg)
    np.testing.assert_allclose(a, T.to_numpy(t), rtol=1e-9, atol=1e-11)


def test_finite_difference_gradients_cfg2_small():
    cfg = dict(backbone_block='resnet', upsampling='spc', scale=4, n_blocks=2, n_filters=4)
    xs = (2, 5, 6, 1)
    P = M.init_params('net_postupsampling', xs, seed=5, dtype=np.float64, **cfg)
    rng = np.random.default_rng(2)
    x = rng.standard_normal(xs)
    y = rng.standard_normal((2, 20, 24, 1))
    PT = M.convert(P, T, requires_grad=True)
    for loss in ('mae', 'mse', 'dssim_mae_mse'):
        lv, grads, _ = TR.supervised_step('net_postupsampling', cfg, PT, T.asarray(x), None,
                                          T.asarray(y), loss=loss)

        def f(Pn):
            pred = M.net_postupsampling(N, Pn, x, None, **cfg)
            return getattr(N, loss)(y, pred)

        assert f(P) == pytest.approx(lv, rel=1e-10)
        for name in ('stem/kernel', 'SubpixelConvolution/conv2x/kernel', 'ConvBlock_att/att/conv1/kernel',
                     'ResidualBlock2/conv1x1/bias', 'ConvBlock_out/conv2/kernel'):
            g = grads[name].numpy()
            idx = np.unravel_index(np.argmax(np.abs(g)), g.shape)
            eps = 1e-6
            Pp = M.Params(); Pp.update({k: np.array(v) for k, v in P.items()})
            Pm = M.Params(); Pm.update({k: np.array(v) for k, v in P.items()})
            Pp[name][idx] += eps
            Pm[name][idx] -= eps
            fd = (f(Pp) - f(Pm)) / (2 * eps)
            assert fd == pytest.approx(g[idx], rel=2e-4, abs=1e-9), (loss, name)


def test_cgan_step_runs_and_discriminator_grad_separation():
    gcfg = dict(n_filters=4, n_blocks=2, decoder_upsampling='dc')
    dcfg = dict(upsampling='pin', scale=8, n_filters=4, n_res_blocks=1)
    PG = M.init_params('unet_pin', (1, 16, 16, 3), (1, 16, 16, 1), seed=1, dtype=np.float64, **gcfg)
    PD = M.Params(create=True, seed=2, dtype=np.float64)
    M.residual_discriminator(N, PD, np.zeros((1, 16, 16, 3)), np.zeros((1, 16, 16, 1)), **dcfg)
    PD.create = False
    rng = np.random.default_rng(0)
    lr = T.asarray(rng.random((2, 16, 16, 3)))
    hr = T.asarray(rng.random((2, 16, 16, 1)))
    st = T.asarray(rng.random((2, 16, 16, 1)))
    out = TR.cgan_step('unet_pin', gcfg, M.convert(PG, T, requires_grad=True), dcfg,
                       M.convert(PD, T, requires_grad=True), lr, hr, st)
    assert out['gen_total'] == pytest.approx(out['gen_gan'] + 100 * out['gen_px'])
    assert all(g is not None for g in out['gradsG'].values())
    assert all(g is not None for g in out['gradsD'].values())


def test_synthetic_batch_shapes():
    x, y = TR.synthetic_batch(1002, 2, 32, 4)
    assert x.shape == (2, 8, 8, 1) and y.shape == (2, 32, 32, 1)
    np.testing.assert_allclose(x[0, 0, 0, 0], y[0, :4, :4, 0].mean(), rtol=1e-5)
