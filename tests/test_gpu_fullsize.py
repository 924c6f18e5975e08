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
