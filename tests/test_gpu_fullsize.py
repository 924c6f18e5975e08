"""Size-independent properties at the BASELINE grid sizes (where the CPU oracle would take minutes per case): batch
independence, bitwise repeatability, exact linearity in the last layer, the closed-form bias gradient of the MAE loss
and a directional-derivative check of the whole backward pass against the forward pass -- cfg2 (4x residual + sub-pixel,
128 -> 512), cfg4 (recurrent, T=8, 64 -> 256) and cfg5's generator (U-Net, 512^2).  The small-grid parity tests
(test_gpu_models.py) pin the arithmetic to the oracle; these pin the tiling / indexing / accumulation at full size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cfg2(seed=3):
    import dl4ds_amd.models as PM
    return PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (128, 128), seed=seed)


def _randomise_biases(model, seed=9):
    rng = np.random.default_rng(seed)
    w = model.get_weights()
    for k in w:
        if k.endswith('bias'):
            w[k] = (rng.standard_normal(w[k].shape) * 0.05).astype(np.float32)
    model.set_weights(w)
    return w


def test_cfg2_full_size_batch_independence_and_repeatability():
    model = _cfg2()
    _randomise_biases(model)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 128, 128, 1)).astype(np.float32)
    y = model([x])
    assert y.shape == (3, 512, 512, 1) and np.isfinite(y).all()
    np.testing.assert_array_equal(y, model([x]))                       # bitwise repeatable
    for i in range(3):                                                 # a sample's output does not depend on its batch
        np.testing.assert_allclose(model([x[i:i + 1]])[0], y[i], rtol=0, atol=2e-6 * np.abs(y).max())
    # permuting the batch permutes the output
    np.testing.assert_allclose(model([x[::-1].copy()]), y[::-1], rtol=0, atol=2e-6 * np.abs(y).max())


def test_cfg2_full_size_last_layer_linearity():
    model = _cfg2()
    w = _randomise_biases(model)
    x = np.random.default_rng(1).standard_normal((2, 128, 128, 1)).astype(np.float32)
    y1 = model([x])
    w2 = dict(w)
    for k in ('ConvBlock_out/conv2/kernel', 'ConvBlock_out/conv2/bias'):
        w2[k] = (w[k] * np.float32(-2.0)).astype(np.float32)           # exact in fp32
    model.set_weights(w2)
    np.testing.assert_array_equal(model([x]), y1 * np.float32(-2.0))


def test_cfg2_full_size_mae_bias_gradient_closed_form():
    from dl4ds_amd.training import SupervisedEngine
    model = _cfg2()
    _randomise_biases(model)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 128, 128, 1)).astype(np.float32)
    y = rng.standard_normal((2, 512, 512, 1)).astype(np.float32)
    pred = model([x])
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    loss, grads = eng.loss_and_grads([x], y)
    assert loss == pytest.approx(float(np.abs(pred.astype(np.float64) - y).mean()), rel=1e-5)
    # L = mean|pred - y|  =>  dL/db_last = sum(sign(pred - y)) / N   (the last conv has no activation)
    expect = float(np.sign(pred.astype(np.float64) - y).sum() / pred.size)
    assert float(grads['ConvBlock_out/conv2/bias'][0]) == pytest.approx(expect, abs=2e-6)
    # and it is bitwise repeatable (deterministic reductions)
    loss2, grads2 = eng.loss_and_grads([x], y)
    assert loss2 == loss
    for k in grads:
        np.testing.assert_array_equal(grads[k], grads2[k], err_msg=k)


def test_cfg2_full_size_directional_derivative():
    """loss(w + e d) - loss(w - e d) = 2 e <grad, d> + O(e^3) for the smooth MSE loss: checks every backward kernel
    against the forward kernels at full size."""
    from dl4ds_amd.training import SupervisedEngine
    model = _cfg2()
    w = _randomise_biases(model)
    rng = np.random.default_rng(3)
    x = rng.standard_normal((2, 128, 128, 1)).astype(np.float32)
    y = rng.standard_normal((2, 512, 512, 1)).astype(np.float32)
    eng = SupervisedEngine(model, loss='mse', learning_rate=1e-3)
    loss0, grads = eng.loss_and_grads([x], y)
    for trial in range(2):
        d = {k: rng.standard_normal(v.shape).astype(np.float32) * (np.abs(v).mean() + 1e-3) for k, v in w.items()}
        gd = sum(float((grads[k].astype(np.float64) * d[k]).sum()) for k in w)
        eps = 2e-3
        ls = []
        for sgn in (+1.0, -1.0):
            model.set_weights({k: (w[k] + np.float32(sgn * eps) * d[k]).astype(np.float32) for k in w})
            ls.append(eng.evaluate([x], y))
        model.set_weights(w)
        fd = (ls[0] - ls[1]) / (2 * eps)
        assert fd == pytest.approx(gd, rel=2e-2, abs=1e-4 * abs(loss0)), (trial, fd, gd)


def test_cfg4_full_size_batch_independence():
    import dl4ds_amd.models as PM
    model = PM.recnet_postupsampling('densenet', 'rc', 4, 1, 1, (64, 64), time_window=8, attention=True,
                                     localcon_layer=True, seed=3)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 8, 64, 64, 1)).astype(np.float32)
    aux = rng.standard_normal((2, 256, 256, 1)).astype(np.float32)
    y = model([x, aux])
    assert y.shape == (2, 8, 256, 256, 1) and np.isfinite(y).all()
    np.testing.assert_array_equal(y, model([x, aux]))
    for i in range(2):
        np.testing.assert_allclose(model([x[i:i + 1], aux[i:i + 1]])[0], y[i], rtol=0, atol=3e-6 * np.abs(y).max())


def test_cfg5_generator_full_size_batch_independence_and_gradient():
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    gen = PM.unet_pin('unet', 5, 1, hr_size=(512, 512), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=3)
    rng = np.random.default_rng(5)
    lr = rng.standard_normal((2, 512, 512, 5)).astype(np.float32)
    st = rng.standard_normal((2, 512, 512, 1)).astype(np.float32)
    y = gen([lr, st])
    assert y.shape == (2, 512, 512, 1) and np.isfinite(y).all()
    np.testing.assert_array_equal(y, gen([lr, st]))
    for i in range(2):
        np.testing.assert_allclose(gen([lr[i:i + 1], st[i:i + 1]])[0], y[i], rtol=0, atol=3e-6 * np.abs(y).max())
    # directional derivative through the 13.6 M-parameter U-Net (9x9 transposed convolutions, max-pooling, skips)
    hr = rng.standard_normal((2, 512, 512, 1)).astype(np.float32)
    w = gen.get_weights()
    eng = SupervisedEngine(gen, loss='mse', learning_rate=1e-3)
    loss0, grads = eng.loss_and_grads([lr, st], hr)
    d = {k: rng.standard_normal(v.shape).astype(np.float32) * (np.abs(v).mean() + 1e-3) for k, v in w.items()}
    gd = sum(float((grads[k].astype(np.float64) * d[k]).sum()) for k in w)
    eps = 1e-3
    ls = []
    for sgn in (+1.0, -1.0):
        gen.set_weights({k: (w[k] + np.float32(sgn * eps) * d[k]).astype(np.float32) for k in w})
        ls.append(eng.evaluate([lr, st], hr))
    fd = (ls[0] - ls[1]) / (2 * eps)
    assert fd == pytest.approx(gd, rel=3e-2, abs=1e-4 * abs(loss0)), (fd, gd)


def test_cfg1_full_size_against_the_oracle():
    """BASELINE configs[0] at its own size (net_pin, residual backbone, 2 channels, 128 x 128, B = 2): forward, MAE loss and
    every gradient against the fp64 torch-CPU oracle -- the oracle needs a few seconds here, so this config is compared
    directly instead of through size-independent properties."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    from oracle import torch_ops as T
    from oracle import models as M
    from oracle import train as TR
    model = PM.net_pin('resnet', 2, 0, hr_size=(128, 128), seed=11)
    assert model.count_params() == 121341
    w = _randomise_biases(model)
    rng = np.random.default_rng(1001)
    x = rng.standard_normal((2, 128, 128, 2)).astype(np.float32)
    y = rng.standard_normal((2, 128, 128, 1)).astype(np.float32)
    P = M.Params()
    for k, v in w.items():
        P[k] = v.astype(np.float64)
    PT = M.convert(P, T, requires_grad=True)
    lv, grads, pred = TR.supervised_step('net_pin', dict(backbone_block='resnet'), PT, T.asarray(x.astype(np.float64)), None,
                                         T.asarray(y.astype(np.float64)), loss='mae')
    out = model([x])
    ref = pred.numpy()
    assert np.abs(out - ref).max() / np.abs(ref).max() < 1e-3          # north_star tolerance; observed ~1e-6
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    l_hip, g_hip = eng.loss_and_grads([x], y)
    assert l_hip == pytest.approx(lv, rel=1e-4)
    gscale = max(float(g.abs().max()) for g in grads.values())
    for k in grads:
        assert np.abs(g_hip[k] - grads[k].numpy()).max() / gscale < 1e-3, k


def test_cfg5_full_size_cgan_step_properties():
    """BASELINE configs[4] at its own size (U-Net(dc) generator 13.6 M parameters + residual discriminator at 512^2): the
    WHOLE CGAN step through dl4ds_cgan_step(apply_update=0) -- D on [real; fake], both backward passes through D, the
    adversarial gradient entering the generator -- checked by (1) batch independence: the losses of a 2-sample step are
    the means of the 1-sample steps (same dropout rows), (2) bitwise repeatability, (3) directional derivatives:
    d(disc loss)/d(theta_D) against D's gradients and d(gen total loss)/d(theta_G) -- which runs through the
    discriminator's input gradient -- against G's gradients."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    H, B = 512, 2
    gen = PM.unet_pin('unet', 5, 1, hr_size=(H, H), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=3)
    disc = PM.residual_discriminator(5, 'pin', False, 8, (H // 8, H // 8), n_filters=8, hr_size=(H, H), seed=4)
    assert gen.count_params() == 13566325 and disc.count_params() == 16177
    rng = np.random.default_rng(1005)
    for m in (gen, disc):
        _randomise_biases(m, seed=int(rng.integers(1 << 30)))
    lr = rng.random((B, H, H, 5)).astype(np.float32)
    st = rng.random((B, H, H, 1)).astype(np.float32)
    hr = rng.random((B, H, H, 1)).astype(np.float32)
    mask = (rng.random((2 * B, 16)) > 0.4).astype(np.float32)            # rows: real 0..B-1, fake 0..B-1
    eng = CGANEngine(gen, disc, loss='mse')                              # smooth pixel loss for the finite differences
    base = eng.step([lr, st], hr, dropout_keep=mask, apply_update=False)
    assert all(np.isfinite(base))
    assert base[0] == pytest.approx(base[1] + 100.0 * base[2], rel=1e-5)
    gg, gd = gen.get_gradients(), disc.get_gradients()
    # (2) repeatable bit for bit, losses and both gradient sets
    again = eng.step([lr, st], hr, dropout_keep=mask, apply_update=False)
    assert again == base
    for a, b in ((gg, gen.get_gradients()), (gd, disc.get_gradients())):
        for k in a:
            np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    # (1) batch independence: every loss is a mean over the batch
    singles = [eng.step([lr[i:i + 1], st[i:i + 1]], hr[i:i + 1], dropout_keep=mask[[i, B + i]], apply_update=False)
               for i in range(B)]
    for j in range(4):
        assert base[j] == pytest.approx(np.mean([s[j] for s in singles]), rel=2e-5), j
    # (3) directional derivatives, one model at a time (the other one's weights stay fixed)
    for model, grads, idx, rel_tol in ((disc, gd, 3, 3e-2), (gen, gg, 0, 3e-2)):
        w = model.get_weights()
        d = {k: rng.standard_normal(v.shape).astype(np.float32) * (np.abs(v).mean() + 1e-3) for k, v in w.items()}
        gdot = sum(float((grads[k].astype(np.float64) * d[k]).sum()) for k in w)
        eps = 1e-3
        ls = []
        for sgn in (+1.0, -1.0):
            model.set_weights({k: (w[k] + np.float32(sgn * eps) * d[k]).astype(np.float32) for k in w})
            ls.append(eng.step([lr, st], hr, dropout_keep=mask, apply_update=False)[idx])
        model.set_weights(w)
        fd = (ls[0] - ls[1]) / (2 * eps)
        assert fd == pytest.approx(gdot, rel=rel_tol, abs=1e-4 * abs(base[idx])), (model.name, fd, gdot)
