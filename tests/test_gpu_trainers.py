"""End-to-end runs of the reference-shaped API on MI355X: SupervisedTrainer.run(), Predictor.run(),
CGANTrainer.run() on tiny synthetic data (loss must go down; shapes/dtypes as the reference returns)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fields(n, hw, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:hw, 0:hw] / hw
    base = np.sin(6 * xx)[None] * np.cos(4 * yy)[None]
    return (base + 0.3 * rng.random((n, 1, 1)) + 0.05 * rng.random((n, hw, hw)))[..., None].astype(np.float32)


def test_supervised_trainer_and_predictor():
    from dl4ds_amd.training import SupervisedTrainer
    from dl4ds_amd.inference import Predictor
    tr, va, te = _fields(24, 32, 0), _fields(8, 32, 1), _fields(8, 32, 2)
    topo = np.random.default_rng(3).random((32, 32)).astype(np.float32)
    t = SupervisedTrainer('resnet', 'spc', tr, va, te, static_vars=[topo], scale=4, batch_size=4, loss='dssim_mae',
                          epochs=4, learning_rate=(2e-3, 1e-4), verbose=False, n_blocks=2, n_filters=4, save=False)
    t.run()
    assert t.model.name == 'resnet_spc'
    assert len(t.fithist['loss']) == 4 and t.fithist['loss'][-1] < t.fithist['loss'][0]
    assert np.isfinite(t.test_loss)
    lr = te.reshape(8, 8, 4, 8, 4, 1).mean(axis=(2, 4))
    y = Predictor(t, lr, scale=4, static_vars=[topo], batch_size=3).run()
    assert y.shape == (8, 32, 32, 1) and y.dtype == np.float32 and np.isfinite(y).all()


def test_supervised_trainer_spatiotemporal_pin():
    from dl4ds_amd.training import SupervisedTrainer
    tr, va, te = _fields(12, 16, 0), _fields(8, 16, 1), _fields(8, 16, 2)
    t = SupervisedTrainer('resnet', 'pin', tr, va, te, scale=2, time_window=3, batch_size=2, epochs=2, verbose=False,
                          n_blocks=1, n_filters=4, learning_rate=1e-3)
    t.run()
    assert t.model.name == 'recresnet_pin' and t.model.output_shape == (3, 16, 16, 1)
    assert np.isfinite(t.fithist['loss']).all()


def test_cgan_trainer_runs():
    from dl4ds_amd.training import CGANTrainer
    tr, te = _fields(16, 32, 0), _fields(4, 32, 1)
    topo = np.random.default_rng(3).random((32, 32)).astype(np.float32)
    t = CGANTrainer('unet', 'pin', tr, te, static_vars=[topo], scale=4, batch_size=4, epochs=2, verbose=False,
                    generator_params=dict(n_filters=4, n_blocks=2, decoder_upsampling='dc'),
                    discriminator_params=dict(n_filters=4, n_res_blocks=1))
    t.run()
    assert len(t.gentotal) == 8 and np.isfinite(t.gentotal).all() and np.isfinite(t.disc).all()
    assert t.generator.name == 'unet_pin'
    assert all(abs(a - (b + 100 * c)) < 1e-3 * abs(a) for a, b, c in zip(t.gentotal, t.gengan, t.genpxloss))
