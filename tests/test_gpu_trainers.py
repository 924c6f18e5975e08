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


def test_supervised_trainer_msdssim_loss():
    """loss='msdssim_mae' through the whole trainer (losses.py:133-139): HR grid 96 px so that all four scales hold
    the 11-tap window; the loss must fall and stay finite."""
    from dl4ds_amd.training import SupervisedTrainer
    tr, va, te = _fields(16, 96, 0), _fields(4, 96, 1), _fields(4, 96, 2)
    t = SupervisedTrainer('resnet', 'spc', tr, va, te, scale=2, batch_size=4, loss='msdssim_mae', epochs=4,
                          learning_rate=2e-3, verbose=False, n_blocks=2, n_filters=8, save=False)
    t.run()
    assert np.isfinite(t.fithist['loss']).all() and t.fithist['loss'][-1] < t.fithist['loss'][0]
    assert np.isfinite(t.test_loss)


def test_supervised_trainer_spatiotemporal_pin():
    from dl4ds_amd.training import SupervisedTrainer
    tr, va, te = _fields(12, 16, 0), _fields(8, 16, 1), _fields(8, 16, 2)
    t = SupervisedTrainer('resnet', 'pin', tr, va, te, scale=2, time_window=3, batch_size=2, epochs=2, verbose=False,
                          n_blocks=1, n_filters=4, learning_rate=1e-3)
    t.run()
    assert t.model.name == 'recresnet_pin' and t.model.output_shape == (3, 16, 16, 1)
    assert np.isfinite(t.fithist['loss']).all()
    # Predictor on a spatio-temporal model: windows are collapsed back into a frame sequence (inference.py:241-242)
    from dl4ds_amd.inference import Predictor
    y = Predictor(t, te, scale=2, array_in_hr=True, time_window=3, batch_size=2).run()
    # n - (time_window - 1) windows (inference.py:187-189) collapse back into exactly n frames (utils.py:32-45)
    assert y.ndim == 4 and y.shape == (te.shape[0], 16, 16, 1) and np.isfinite(y).all()


def test_predictor_on_pin_models_with_lr_input():
    """array_in_hr=False on a 'pin' model -- the LR array is first re-expanded to the HR grid and handed over
    as array_lr (inference.py:196-203); the prediction must equal the one obtained from the HR-side construction of the
    same inputs."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.inference import Predictor, predict
    hr = _fields(6, 32, 5)
    lr = hr.reshape(6, 8, 4, 8, 4, 1).mean(axis=(2, 4)).astype(np.float32)
    for model in (PM.net_pin('resnet', 1, 0, hr_size=(32, 32), n_blocks=2, n_filters=4, seed=1),
                  PM.unet_pin('unet', 1, 0, hr_size=(32, 32), n_blocks=2, n_filters=4, seed=2)):
        y_lr = Predictor(model, lr, scale=4, batch_size=4).run()                       # array_in_hr=False (class default)
        assert y_lr.shape == (6, 32, 32, 1) and np.isfinite(y_lr).all()
        # the same model input built by hand: block means replicated back to the HR grid
        x = np.repeat(np.repeat(lr, 4, axis=1), 4, axis=2)
        np.testing.assert_allclose(y_lr, model.predict([x], batch_size=3), rtol=1e-5, atol=1e-6)
        y_hr, x_used = predict(model, hr, scale=4, return_lr=True)                      # array_in_hr=True (function default)
        np.testing.assert_allclose(x_used, x, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(y_hr, y_lr, rtol=1e-4, atol=1e-5)


def test_cgan_learning_rates_forms():
    """cgan.py:271-278: a (genlr, dislr) pair gives the two optimisers their own rates; a float or a 1-tuple sets both."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    from dl4ds_amd.training.engine import cgan_learning_rates
    assert cgan_learning_rates(2e-4) == (2e-4, 2e-4) and cgan_learning_rates((1e-3,)) == (1e-3, 1e-3)
    assert cgan_learning_rates([2e-4, 1e-4]) == (2e-4, 1e-4)
    with pytest.raises(TypeError):
        cgan_learning_rates('fast')
    rng = np.random.default_rng(0)
    B, H = 2, 16
    lr, st, hr = (rng.random((B, H, H, c)).astype(np.float32) for c in (2, 1, 1))
    mask = (rng.random((2 * B, 8)) > 0.4).astype(np.float32)

    def one_step(rates):
        gen = PM.unet_pin('unet', 2, 1, hr_size=(H, H), n_filters=4, n_blocks=2, decoder_upsampling='dc', seed=3)
        disc = PM.residual_discriminator(2, 'pin', False, 8, (H // 8, H // 8), n_filters=4, n_res_blocks=1, hr_size=(H, H), seed=4)
        g0, d0 = gen.get_weights(), disc.get_weights()
        CGANEngine(gen, disc, loss='mae', learning_rate=rates).step([lr, st], hr, dropout_keep=mask)
        dg = max(float(np.abs(gen.get_weights()[k] - g0[k]).max()) for k in g0)
        dd = max(float(np.abs(disc.get_weights()[k] - d0[k]).max()) for k in d0)
        return dg, dd

    # the first Adam step moves a weight by at most lr (|m| / sqrt(v) = 1): the step sizes expose the two rates
    dg, dd = one_step((4e-4, 1e-4))
    assert 3.9e-4 < dg < 4.01e-4 and 0.97e-4 < dd < 1.01e-4, (dg, dd)
    dg, dd = one_step(3e-4)
    assert 2.9e-4 < dg < 3.01e-4 and 2.9e-4 < dd < 3.01e-4, (dg, dd)


def test_supervised_trainer_early_stopping_and_best_model_files(tmp_path):
    """EarlyStopping(monitor='val_loss', patience) and ModelCheckpoint(best_model) (supervised.py:356-390)."""
    import os
    from dl4ds_amd.training import SupervisedTrainer
    tr, va, te = _fields(8, 16, 0), _fields(4, 16, 1), _fields(4, 16, 2)
    t = SupervisedTrainer('resnet', 'spc', tr, va, te, scale=2, batch_size=4, epochs=6, learning_rate=1e-3, verbose=False,
                          n_blocks=1, n_filters=4, save=True, save_path=str(tmp_path), save_bestmodel=True,
                          early_stopping=True, patience=2, min_delta=10.0)        # nothing improves by 10: stops after 1 + 2 epochs
    t.run()
    assert len(t.fithist['val_loss']) == 3
    files = set(os.listdir(tmp_path / 'best_model'))
    assert {'checkpoint.npz', 'model_weights.npz', 'epoch_val_loss.txt'} <= files
    ep, vl = np.loadtxt(tmp_path / 'best_model' / 'epoch_val_loss.txt')
    assert int(ep) == 3 and abs(vl - t.fithist['val_loss'][-1]) < 1e-6
    w = np.load(tmp_path / 'best_model' / 'model_weights.npz')
    for k, v in t.model.get_weights().items():
        np.testing.assert_array_equal(w[k], v)


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
    assert np.isfinite(t.test_loss) and t.test_loss > 0          # pixel loss of generator.predict on the test set (cgan.py:386-440)


def test_cgan_trainer_postupsampling_and_spatiotemporal():
    """CGANTrainer with a post-upsampling generator (discriminator HR branch reduced by stride-2 convolutions,
    discriminator.py:52-60), with normalised / dropout generator blocks, and the spatio-temporal pair (:31-33,73-74)."""
    from dl4ds_amd.training import CGANTrainer
    tr, te = _fields(16, 32, 0), _fields(6, 32, 1)
    topo = np.random.default_rng(3).random((32, 32)).astype(np.float32)
    t = CGANTrainer('resnet', 'spc', tr, te, static_vars=[topo], scale=4, batch_size=4, epochs=2, verbose=False,
                    generator_params=dict(n_filters=4, n_blocks=1, normalization='ln', dropout_rate=0.1),
                    discriminator_params=dict(n_filters=4, n_res_blocks=1, normalization='ln'))
    t.run()
    assert t.generator.name == 'resnet_spc' and len(t.gentotal) == 8
    assert np.isfinite(t.gentotal).all() and np.isfinite(t.disc).all()
    t = CGANTrainer('convnet', 'pin', tr, te, static_vars=[topo], scale=2, time_window=2, batch_size=2, epochs=1,
                    verbose=False, generator_params=dict(n_filters=4, n_blocks=1),
                    discriminator_params=dict(n_filters=4, n_res_blocks=1))
    t.run()
    assert t.generator.name == 'recconvnet_pin'
    assert np.isfinite(t.gentotal).all() and np.isfinite(t.disc).all()


def test_supervised_trainer_batchnorm_dropout():
    """normalization='bn' + Gaussian dropout through SupervisedTrainer: training uses batch statistics and noise, the
    validation / test losses and Predictor run in inference mode on the moving statistics."""
    from dl4ds_amd.training import SupervisedTrainer
    from dl4ds_amd.inference import Predictor
    tr, va, te = _fields(24, 32, 0), _fields(8, 32, 1), _fields(8, 32, 2)
    t = SupervisedTrainer('resnet', 'spc', tr, va, te, scale=4, batch_size=4, epochs=3, learning_rate=2e-3, verbose=False,
                          n_blocks=2, n_filters=4, normalization='bn', dropout_rate=0.1, dropout_variant='gaussian',
                          save=False)
    t.run()
    assert np.isfinite(t.fithist['loss']).all() and np.isfinite(t.fithist['val_loss']).all() and np.isfinite(t.test_loss)
    w = t.model.get_weights()
    mv = [k for k in w if k.endswith('moving_variance')]
    assert mv and any(not np.allclose(w[k], 1.0) for k in mv)                 # moving statistics were maintained
    assert len(t.model.trainable_variables) == len(w) - 2 * len(mv)
    lr = te.reshape(8, 8, 4, 8, 4, 1).mean(axis=(2, 4))
    y1 = Predictor(t, lr, scale=4, batch_size=3).run()
    y2 = Predictor(t, lr, scale=4, batch_size=8).run()
    np.testing.assert_allclose(y1, y2, rtol=1e-5, atol=1e-6)                   # inference does not depend on the batch


def test_bucketed_rccl_allreduce_single_rank_matches_local_step():
    """The data-parallel step (gradient buckets all-reduced on the communication stream while the backward pass is
    still running, 1/world folded into Adam) on a 1-rank RCCL communicator must reproduce the plain step bit for bit:
    sum over one rank and 1/1 are exact.  Run in a child process so the communicator does not leak into other tests."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        import dl4ds_amd.models as PM
        from dl4ds_amd.training import SupervisedEngine
        from dl4ds_amd import parallel
        rng = np.random.default_rng(0)
        x = rng.standard_normal((2, 16, 16, 1)).astype(np.float32)
        y = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)
        def run(dist):
            m = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (16, 16), n_blocks=3, seed=5)
            e = SupervisedEngine(m, loss='mae', learning_rate=1e-3)
            if dist:
                parallel.broadcast_trainer(e)
            losses = [e.step([x], y) for _ in range(3)]
            return losses, m.get_weights()
        l0, w0 = run(False)
        parallel.init_with_id(0, 1, parallel.unique_id())
        l1, w1 = run(True)
        parallel.finalize()
        assert l0 == l1, (l0, l1)
        for k in w0:
            np.testing.assert_array_equal(w0[k], w1[k], err_msg=k)
        print('BUCKETS-OK')
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert 'BUCKETS-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_cgan_overlapped_allreduce_single_rank_matches_local_step():
    """CGAN data-parallel step: the discriminator's gradients are all-reduced on the communication stream underneath
    the generator pass, the generator's buckets as its backward completes them; on a 1-rank RCCL communicator (sum over
    one rank, 1/1) three steps must reproduce the local steps bit for bit -- losses and both weight sets."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        import dl4ds_amd.models as PM
        from dl4ds_amd.training import CGANEngine
        from dl4ds_amd import parallel
        rng = np.random.default_rng(0)
        B, H = 2, 32
        lr = rng.random((B, H, H, 2)).astype(np.float32)
        st = rng.random((B, H, H, 1)).astype(np.float32)
        hr = rng.random((B, H, H, 1)).astype(np.float32)
        mask = (rng.random((2 * B, 8)) > 0.4).astype(np.float32)
        def run():
            gen = PM.unet_pin('unet', 2, 1, hr_size=(H, H), n_filters=4, n_blocks=3, decoder_upsampling='dc', seed=3)
            disc = PM.residual_discriminator(2, 'pin', False, 8, (H // 8, H // 8), n_filters=4, n_res_blocks=2,
                                             hr_size=(H, H), seed=4)
            e = CGANEngine(gen, disc, loss='mae')
            losses = [e.step([lr, st], hr, dropout_keep=mask) for _ in range(3)]
            return losses, gen.get_weights(), disc.get_weights()
        l0, g0, d0 = run()
        parallel.init_with_id(0, 1, parallel.unique_id())
        l1, g1, d1 = run()
        parallel.finalize()
        assert l0 == l1, (l0, l1)
        for a, b in ((g0, g1), (d0, d1)):
            for k in a:
                np.testing.assert_array_equal(a[k], b[k], err_msg=k)
        print('CGAN-DP-OK')
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert 'CGAN-DP-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_checkpoint_resume_continues_the_same_optimisation(tmp_path):
    """Weights + Adam slots + iteration count round-trip through SupervisedEngine.save_checkpoint / load_checkpoint:
    3 + 3 steps with a restore in between equal 6 uninterrupted steps bit for bit."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 16, 16, 1)).astype(np.float32)
    y = rng.standard_normal((2, 64, 64, 1)).astype(np.float32)

    def fresh():
        m = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (16, 16), n_blocks=2, seed=5)
        return m, SupervisedEngine(m, loss='mae', learning_rate=(1e-3, 1e-4), lr_decay_after=4)

    m_a, e_a = fresh()
    ref_losses = [e_a.step([x], y) for _ in range(6)]
    m_b, e_b = fresh()
    first = [e_b.step([x], y) for _ in range(3)]
    e_b.save_checkpoint(tmp_path / 'ck.npz')
    m_c, e_c = fresh()
    m_c.set_weights({k: np.zeros_like(v) for k, v in m_c.get_weights().items()})     # must all be overwritten
    assert e_c.load_checkpoint(tmp_path / 'ck.npz') == 3
    second = [e_c.step([x], y) for _ in range(3)]
    assert first + second == ref_losses
    for k, v in m_a.get_weights().items():
        np.testing.assert_array_equal(m_c.get_weights()[k], v, err_msg=k)
    assert e_c.optimizer_state()[2] == 6


def test_cgan_checkpoint_resume_and_trainer_files(tmp_path):
    """CGANEngine.save_checkpoint / load_checkpoint (both models, both Adam states; cgan.py:288-292,447-522): 2 + 2 steps
    with a restore in between equal 4 uninterrupted steps bit for bit; CGANTrainer(checkpoints_frequency=1) writes the
    per-epoch and final files (cgan.py:370-382)."""
    import os
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine, CGANTrainer
    rng = np.random.default_rng(0)
    B, H = 2, 16
    lr, st, hr = (rng.random((B, H, H, c)).astype(np.float32) for c in (2, 1, 1))
    mask = (rng.random((2 * B, 8)) > 0.4).astype(np.float32)

    def fresh():
        gen = PM.unet_pin('unet', 2, 1, hr_size=(H, H), n_filters=4, n_blocks=2, decoder_upsampling='dc', seed=3)
        disc = PM.residual_discriminator(2, 'pin', False, 8, (H // 8, H // 8), n_filters=4, n_res_blocks=1, hr_size=(H, H), seed=4)
        return gen, disc, CGANEngine(gen, disc, loss='mae')

    g_a, d_a, e_a = fresh()
    ref = [e_a.step([lr, st], hr, dropout_keep=mask) for _ in range(4)]
    g_b, d_b, e_b = fresh()
    first = [e_b.step([lr, st], hr, dropout_keep=mask) for _ in range(2)]
    e_b.save_checkpoint(str(tmp_path / 'ck.npz'))
    g_c, d_c, e_c = fresh()
    e_c.load_checkpoint(str(tmp_path / 'ck.npz'))
    assert e_c.optimizer_state('generator')[2] == 2 and e_c.optimizer_state('discriminator')[2] == 2
    rest = [e_c.step([lr, st], hr, dropout_keep=mask) for _ in range(2)]
    assert first + rest == ref
    for a, b in ((g_a, g_c), (d_a, d_c)):
        wa, wb = a.get_weights(), b.get_weights()
        for k in wa:
            np.testing.assert_array_equal(wa[k], wb[k], err_msg=k)
    tr, te = _fields(8, 32, 0), _fields(4, 32, 1)
    topo = rng.random((32, 32)).astype(np.float32)
    t = CGANTrainer('unet', 'pin', tr, te, static_vars=[topo], scale=4, batch_size=4, epochs=2, verbose=False,
                    checkpoints_frequency=1, save_path=str(tmp_path) + '/',
                    generator_params=dict(n_filters=4, n_blocks=2, decoder_upsampling='dc'),
                    discriminator_params=dict(n_filters=4, n_res_blocks=1))
    t.run()
    files = sorted(os.listdir(tmp_path / 'checkpoints'))
    assert 'checkpoint_epoch-1.npz' in files and 'checkpoint_epoch-2.npz' in files and 'save_epoch2_generator_weights.npz' in files


def test_cgan_module_level_functions(tmp_path):
    """training/cgan.py:447-639: generator_loss, discriminator_loss, train_step and load_checkpoint under their own names."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import cgan as C
    from dl4ds_amd import losses
    rng = np.random.default_rng(0)
    B, H = 2, 16
    d_real, d_fake = rng.random((B, 1)).astype(np.float32) * 0.8 + 0.1, rng.random((B, 1)).astype(np.float32) * 0.8 + 0.1
    gen_out, target = rng.random((B, H, H, 1)).astype(np.float32), rng.random((B, H, H, 1)).astype(np.float32)
    tot, gan, px = C.generator_loss(d_fake, gen_out, target, losses.mae)
    assert gan == pytest.approx(float(-np.log(d_fake).mean()), rel=1e-5)
    assert px == pytest.approx(float(np.abs(gen_out - target).mean()), rel=1e-5) and tot == pytest.approx(gan + 100 * px, rel=1e-6)
    tot10 = C.generator_loss(d_fake, gen_out, target, 'mse', lambda_scaling_factor=10)[0]
    assert tot10 == pytest.approx(gan + 10 * float(((gen_out - target) ** 2).mean()), rel=1e-5)
    dl = C.discriminator_loss(d_real, d_fake)
    assert dl == pytest.approx(float(-np.log(d_real).mean() - np.log(1 - d_fake).mean()), rel=1e-5)
    # train_step with optimiser handles; checkpoint written by CGANTrainer restored by load_checkpoint
    tr, te = _fields(8, 16, 0), _fields(4, 16, 1)
    topo = rng.random((16, 16)).astype(np.float32)
    t = C.CGANTrainer('resnet', 'pin', tr, te, static_vars=[topo], scale=2, batch_size=4, epochs=1, verbose=False,
                      checkpoints_frequency=1, save_path=str(tmp_path) + '/',
                      generator_params=dict(n_filters=4, n_blocks=2), discriminator_params=dict(n_filters=4, n_res_blocks=1))
    t.run()
    g, gopt, d, dopt = C.load_checkpoint(str(tmp_path / 'checkpoints'), 1, 'resnet', 'pin', 2, (16, 16), n_static_vars=1,
                                         n_blocks=(2, 1), n_filters=(4, 4))
    for k, v in t.generator.get_weights().items():
        np.testing.assert_array_equal(g.get_weights()[k], v)
    for k, v in t.discriminator.get_weights().items():
        np.testing.assert_array_equal(d.get_weights()[k], v)
    assert gopt.iterations == dopt.iterations == t.engine.optimizer_state('generator')[2] == 2
    lr_b = rng.random((4, 16, 16, 2)).astype(np.float32)
    st_b = rng.random((4, 16, 16, 1)).astype(np.float32)
    hr_b = rng.random((4, 16, 16, 1)).astype(np.float32)
    out = C.train_step(lr_b, hr_b, g, d, gopt, dopt, 0, losses.mae, None, True, static_array=st_b)
    assert len(out) == 4 and all(np.isfinite(out)) and gopt.iterations == 3
    with pytest.raises(ValueError):
        C.train_step(lr_b, hr_b, t.generator, d, gopt, dopt, 0, 'mae', None, False, static_array=st_b)


def test_msdssim_loss_callables_exist_and_agree_with_the_kernels():
    from dl4ds_amd import losses, ops
    rng = np.random.default_rng(1)
    a, b = rng.random((2, 96, 96, 1)).astype(np.float32), rng.random((2, 96, 96, 1)).astype(np.float32)
    for name in ('msdssim', 'msdssim_mae', 'msdssim_mae_mse', 'dssim', 'mae'):
        f = getattr(losses, name)
        assert f.__name__ == name
        assert f(a, b) == ops.loss(name, a, b, want_grad=False)[0]
    assert losses.msdssim(a, a) == pytest.approx(0.0, abs=1e-6)


def test_predict_on_a_grid_other_than_the_one_the_model_was_built_for():
    """The reference generators take Input(shape=(None, None, C)) (sp_postups.py:112-115): a trained model predicts on any
    grid.  Model.predict re-plans the graph for the new grid with the current weights -- the result must equal a model
    built for that grid directly, and follow later weight changes."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.inference import Predictor
    rng = np.random.default_rng(2)
    m = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (16, 16), n_blocks=2, n_filters=8, seed=1)
    w = m.get_weights()
    for k in w:
        if k.endswith('bias'):
            w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    m.set_weights(w)
    x = rng.standard_normal((3, 24, 20, 1)).astype(np.float32)
    direct = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (24, 20), n_blocks=2, n_filters=8, seed=9)
    direct.set_weights(w)
    y = m.predict(x, batch_size=2)
    assert y.shape == (3, 96, 80, 1)
    np.testing.assert_array_equal(y, direct.predict(x, batch_size=2))
    np.testing.assert_array_equal(Predictor(m, x, scale=4, batch_size=3).run(), y)       # LR array in, other domain size
    w2 = {k: (v * np.float32(0.5)).astype(np.float32) for k, v in w.items()}
    m.set_weights(w2); direct.set_weights(w2)
    np.testing.assert_array_equal(m.predict(x), direct.predict(x))                        # cached sibling follows the weights
    # ... also when the weights change on the DEVICE (an optimiser step), and the cache of re-planned graphs stays bounded
    from dl4ds_amd.training import SupervisedEngine
    SupervisedEngine(m, loss='mae', learning_rate=1e-2).step([rng.standard_normal((2, 16, 16, 1)).astype(np.float32)],
                                                             rng.standard_normal((2, 64, 64, 1)).astype(np.float32))
    direct.set_weights(m.get_weights())
    np.testing.assert_array_equal(m.predict(x), direct.predict(x))
    for g in ((8, 8), (9, 12), (12, 9), (10, 10), (11, 8)):
        assert m.predict(rng.standard_normal((1,) + g + (1,)).astype(np.float32)).shape == (1, 4 * g[0], 4 * g[1], 1)
    assert len(m._resized_cache) == m.RESIZED_CACHE_GRIDS == 4 and (24, 20) not in m._resized_cache
    pin = PM.unet_pin('unet', 2, 1, hr_size=(32, 32), n_filters=4, n_blocks=2, seed=3)
    xs = [rng.standard_normal((2, 48, 40, 2)).astype(np.float32), rng.standard_normal((2, 48, 40, 1)).astype(np.float32)]
    assert pin.predict(xs).shape == (2, 48, 40, 1)
    lcb = PM.net_postupsampling('resnet', 'spc', 2, 1, 0, (16, 16), n_blocks=1, localcon_layer=True, seed=1)
    with pytest.raises(ValueError, match='localcon_layer'):
        lcb.predict(rng.standard_normal((1, 20, 20, 1)).astype(np.float32))


@pytest.mark.parametrize('ups,interp', [('spc', 'bicubic'), ('pin', 'bilinear'), ('rc', 'lanczos')])
def test_supervised_trainer_other_interpolations_prepare_batches_on_the_device(ups, interp):
    """`interpolation` other than the default: the trainers still gather their batches on the device (cv2 tap tables,
    `dl4ds_batch_prepare_taps`), the batches equal the host loader's, and training runs."""
    from dl4ds_amd.training import SupervisedTrainer
    from dl4ds_amd.dataloader import DataGenerator, DeviceDataGenerator
    tr, va, te = _fields(16, 32, 0), _fields(8, 32, 1), _fields(8, 32, 2)
    t = SupervisedTrainer('resnet', ups, tr, va, te, scale=2, interpolation=interp, batch_size=4, epochs=3, patch_size=16,
                          learning_rate=2e-3, verbose=False, n_blocks=1, n_filters=4, save=False)
    t.run()
    assert isinstance(t.ds_train, DeviceDataGenerator) and t.ds_train.taps
    assert np.isfinite(t.fithist['loss']).all() and t.fithist['loss'][-1] < t.fithist['loss'][0]
    host = DataGenerator(tr, None, backbone='resnet', upsampling=ups, scale=2, batch_size=4, patch_size=16,
                         interpolation=interp, seed=5)
    dev = DeviceDataGenerator(tr, None, backbone='resnet', upsampling=ups, scale=2, batch_size=4, patch_size=16,
                              interpolation=interp, seed=5)
    (xh,), (yh,) = host[0]
    (xd,), (yd,) = dev[0]
    np.testing.assert_allclose(xd.numpy(), xh, rtol=0, atol=1e-5 * max(np.abs(xh).max(), 1.0))
    np.testing.assert_array_equal(yd.numpy(), yh)
