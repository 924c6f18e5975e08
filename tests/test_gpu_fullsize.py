"""BASELINE-size checks of the kernels the bench dispatches.

Two kinds of test at the full grid sizes of BASELINE.json's configs:

* DIRECT comparisons with the fp64 torch oracle (``test_cfg*_full_size_against_the_oracle`` and the CGAN step): forward,
  loss and EVERY parameter gradient, each gradient tensor at its own scale (tests/parity.py), at a batch size large enough
  that the producer / consumer kernels of the bench are the ones dispatched -- asserted through the library profiler's
  kernel tags, with no DL4DS_*FORCE* override in the environment.  The oracle is evaluated sample by sample (the losses
  are batch means, the models carry no batch statistics), about 1-3 s of host time per sample.
* size-independent properties (batch independence and permutation, bitwise repeatability, exact linearity in the last
  layer, the closed-form MAE bias gradient, directional derivatives of the whole backward pass) -- these pin the
  tiling / indexing / accumulation without any oracle."""
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
    """BASELINE configs[0] at its own size (net_pin, residual backbone, 2 channels, 128 x 128, B = 8): forward, MAE loss and
    every gradient -- each tensor at its own scale, tests/parity.py -- against the fp64 torch-CPU oracle."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, oracle_reference
    model = PM.net_pin('resnet', 2, 0, hr_size=(128, 128), seed=11)
    assert model.count_params() == 121341
    w = _randomise_biases(model)
    rng = np.random.default_rng(1001)
    x = rng.standard_normal((8, 128, 128, 2)).astype(np.float32)
    out = model([x])
    y = _targets_clear_of_the_kink(out, rng)
    ref = oracle_reference('supervised', 'net_pin', dict(backbone_block='resnet'), w, x, None, y, loss='mae', workers=8)
    assert np.abs(out - ref['pred']).max() / np.abs(ref['pred']).max() < 1e-3          # north_star tolerance; observed ~1e-6
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    l_hip, g_hip = eng.loss_and_grads([x], y)
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    # (8 x 128^2 pixels per gradient entry: the fp32 noise floor of the smallest tensors is 1.9 % here, 0.1-0.4 % at cfg2 / 4 / 5)
    _slack_is_small(assert_matches_reference(g_hip, ref, what='cfg1 full size', full=True), limit=0.03)


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


# ------------------------------------------------------------------------------------------------ direct oracle comparisons
ORACLE_WORKERS = 8         # worker processes of the fp64 oracle (tests/oracle_worker.py), 16 threads each at most


def _no_force_overrides():
    import os
    bad = [k for k in os.environ if k.startswith('DL4DS_') and ('FORCE' in k or k.startswith('DL4DS_NO_'))]
    assert not bad, f'kernel-selection overrides in the environment: {bad}'


def _targets_clear_of_the_kink(pred, rng):
    from tests.parity import targets_clear_of_the_kink          # (oracle/reference.py: shared with __graft_entry__.smoke())
    return targets_clear_of_the_kink(pred, rng)


def _fwd_close(out, ref, tol=1e-3):
    assert out.shape == ref.shape
    err = np.abs(out.astype(np.float64) - ref).max() / np.abs(ref).max()
    assert err < tol, err
    return err


SLACK_LIMIT = 0.01         # largest band + noise floor the oracle may GRANT any entry of a tensor, as a fraction of the tensor's own size
PLAIN_FRACTION = 0.95      # share of the tensors that must pass on north_star's 1e-3 (+ the ulp floor) ALONE


def _slack_is_small(rows, limit=SLACK_LIMIT):
    """``rows`` = tests.parity.assert_matches_reference's per-tensor breakdown (which has already asserted the per-element
    criterion and the element-level caps every comparison is held to: tests/parity.py).  At the BASELINE sizes two more
    statements hold and are asserted:
    (1) at least PLAIN_FRACTION of the TENSORS pass with every entry on tol = 1e-3 + the ulp floor alone
        (realised in round 4: 56/56 cfg2 either kernel set, 58/58 cfg4, 76/76 + 48/48 cfg5, cfg1 see the artefact);
    (2) the largest allowance the oracle grants ANY entry stays a correction: no tensor -- none exempted by name -- holds an entry
        with more than ``limit`` = 1 % of the tensor's size (the two small-sample cases pass limit = 3 % at their call sites)."""
    from tests.parity import summarize
    sm = summarize(rows)
    assert sm['plain_frac'] >= PLAIN_FRACTION, f'only {sm["plain_ok"]} of {sm["n"]} tensors pass on 1e-3 alone: {sm}'
    for r in rows:
        assert r['band'] + r['noise'] < limit, f'oracle slack too large, the comparison has lost its teeth: {r}'
    return sm


_CFG2_REF = {}


@pytest.mark.parametrize('winograd', [True, False, 'split'])
def test_cfg2_full_size_against_the_oracle(monkeypatch, winograd):
    """BASELINE configs[1] (the headline: resnet + sub-pixel x4, 128 -> 512) at B = 32 -- the smallest batch at which every
    producer / consumer kernel of the B = 64 bench step is dispatched (asserted) -- against the fp64 oracle: forward, MAE
    loss, every gradient per tensor; once with the Winograd form of the MFMA-bound 3x3 layers (what the bench runs) and once
    with the direct kernels (DL4DS_NO_WINOGRAD=1), both with DL4DS_NO_SPLIT=1.  Round 6, 'split': the same comparison, UNCHANGED, in the
    product's DEFAULT dispatch -- the eight single-pass <= 48-channel <3,3> layers on the six-term bf16 kernel (conv_split), 48 -> 192 and
    192 -> 48 on the Winograd kernel -- the condition VERDICT r5 set for calling that arithmetic fp32.  sp_postups.py:95-217."""
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, kernel_tags, oracle_reference
    _no_force_overrides()
    split = winograd == 'split'
    monkeypatch.delenv('DL4DS_SPLIT', raising=False)
    monkeypatch.delenv('DL4DS_NO_SPLIT', raising=False)
    if winograd:
        monkeypatch.delenv('DL4DS_NO_WINOGRAD', raising=False)
    else:
        monkeypatch.setenv('DL4DS_NO_WINOGRAD', '1')
    if not split:
        monkeypatch.setenv('DL4DS_NO_SPLIT', '1')          # (True / False: the fp32-pipe kernels alone; 'split': the product's default dispatch)
    B = 32
    model = _cfg2(seed=11)
    w = _randomise_biases(model)
    rng = np.random.default_rng(1002)
    x = rng.standard_normal((B, 128, 128, 1)).astype(np.float32)
    out = model([x])
    y = _CFG2_REF['y'] if 'y' in _CFG2_REF else _targets_clear_of_the_kink(out, rng)
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([x], y))
    # the kernels that carry the bench step (profiles/kernel_stats_r03.txt) are the ones that just ran
    must = ('conv_wino<3,3>', 'conv_wino<3,2>', 'conv_wino<2,3>', 'conv_wino<2,2>', 'conv_wino_wgrad<3,3>', 'conv_wino_wgrad<3,2>',
            'conv_wino_wgrad<2,2>') if winograd else \
           ('conv_stream_ws<3,6,3,8>', 'conv_stream_ws<3,12,2,4>', 'conv_stream_ws<3,8,3,4>', 'conv_wgrad_rows<3,3,1,4>',
            'conv_wgrad_rows<3,3,1,1>', 'conv_wgrad_rows<3,3,1,2>')
    for m in must + ('conv_narrow_pair_ws<4>', 'conv_narrow_wgrad<8>'):
        assert m in tags, (m, sorted(tags))
    assert winograd or not any(t.startswith('conv_wino') for t in tags), sorted(tags)
    assert ('conv_split<3,3>' in tags) == split, sorted(tags)
    if split:
        assert tags['conv_split<3,3>'] == 8 and tags['conv_wino<3,3>'] == 2, (tags['conv_split<3,3>'], tags['conv_wino<3,3>'])
    if 'ref' not in _CFG2_REF:          # (same weights, inputs and targets in both runs: the oracle is evaluated once)
        _CFG2_REF['y'] = y
        _CFG2_REF['ref'] = oracle_reference('supervised', 'net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4),
                                            w, x, None, y, loss='mae', workers=ORACLE_WORKERS)
    ref = _CFG2_REF['ref']
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    _slack_is_small(assert_matches_reference(g_hip, ref, what=f'cfg2 full size ({"default dispatch: six-term bf16 + Winograd" if split else "Winograd" if winograd else "direct"} kernels)', full=True))


@pytest.mark.parametrize('B', [4, 8, 16])
def test_cfg2_on_the_bench_workload_itself_against_the_oracle(B):
    """(Round 6: from B = 8 on the product's DEFAULT dispatch puts the eight single-pass <= 48-channel 3x3 layers on conv_split_kernel --
    asserted: 16-row strips at B = 8, 32-row strips at B = 16 -- and B = 4 runs them on the fp32 pipe, so that both arithmetics are compared
    on data nobody steered.)
    VERDICT r4 weak #3: the comparisons above feed N(0, 1) inputs and targets placed clear of the MAE kink; `bench.py` times something
    else -- box-blurred U[0, 1) fields, LR = their 4 x 4 block means, the HR fields themselves as targets, glorot kernels with the ZERO
    biases the builders start from.  The same step on exactly that data and those weights (bench.synthetic_batch(1002, B), seed 7; B = 16:
    the oracle's cost) against the fp64 oracle: forward, MAE loss and every gradient entry, held to the same caps.  Nothing is steered
    away from a discontinuity here: MAE residuals that sit within the band of their sign change get their own per-entry allowance."""
    import bench
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, kernel_tags, oracle_reference
    _no_force_overrides()
    model = _cfg2(seed=7)
    w = model.get_weights()
    assert all(np.abs(v).max() == 0.0 for k, v in w.items() if k.endswith('bias'))
    x, y = bench.synthetic_batch(1002, B)
    out = model([x])
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([x], y))
    assert any(t.startswith('conv_wino<3,3>') for t in tags) and 'conv_narrow_pair_ws<4>' in tags, sorted(tags)
    assert (tags.get('conv_split<3,3>') == 8) == (B >= 8), sorted(tags)
    ref = oracle_reference('supervised', 'net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4), w, x, None, y,
                           loss='mae', workers=ORACLE_WORKERS)
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    _slack_is_small(assert_matches_reference(g_hip, ref, what=f'cfg2 on the bench workload (box-blurred fields, zero biases, B = {B})', full=True),
                    limit=0.03)


def _cfg4_model(seed=3):
    import dl4ds_amd.models as PM
    return PM.recnet_postupsampling('densenet', 'rc', 4, 1, 1, (64, 64), time_window=8, attention=True,
                                    localcon_layer=True, seed=seed)


CFG4_OCFG = dict(backbone_block='densenet', upsampling='rc', scale=4, time_window=8, attention=True, localcon_layer=True)


def test_cfg4_full_size_against_the_oracle():
    """BASELINE configs[3] (recurrent dense backbone + attention + LCB, resize-convolution x4, T = 8, 64 -> 256) against the
    fp64 oracle.  spt_postups.py:96-163."""
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, kernel_tags, oracle_reference
    _no_force_overrides()
    B = 8
    model = _cfg4_model()
    assert model.count_params() == 480056
    w = _randomise_biases(model)
    rng = np.random.default_rng(1004)
    x = rng.standard_normal((B, 8, 64, 64, 1)).astype(np.float32)
    aux = rng.standard_normal((B, 256, 256, 1)).astype(np.float32)
    out = model([x, aux])
    y = _targets_clear_of_the_kink(out, rng)
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([x, aux], y))
    # the bench's kernels: the persistent ConvLSTM recurrence (both directions, 5x5 and 3x3) and the 16-channel layer
    for must in ('convlstm_seq_fwd<5,8>', 'convlstm_seq_fwd<3,8>', 'convlstm_seq_bwd<5,8>', 'convlstm_seq_bwd<3,8>', 'conv_narrow16_ws<4>'):
        assert must in tags, (must, sorted(tags))
    assert not any(t.startswith('convlstm_gates') for t in tags), sorted(tags)
    ref = oracle_reference('supervised', 'recnet_postupsampling', CFG4_OCFG, w, x, aux, y, loss='mae', workers=ORACLE_WORKERS)
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    # (LocalizedConvBlock's per-grid-point variables sum only B x T = 64 terms per entry; with targets clear of the MAE kink the
    #  oracle grants them 3e-4 of their size -- profiles/parity_r04.json -- so they need no exemption from the cap any more)
    _slack_is_small(assert_matches_reference(g_hip, ref, what='cfg4 full size', full=True))


def test_cfg4_on_the_bench_workload_itself_against_the_oracle():
    """`bench.py --config cfg4` times box-blurred U[0, 1) frames, their block means, a blurred auxiliary field and
    the zero biases the builders start from (bench.synthetic_batch_cfg4(1004, B), model seed 7) -- that step, with nothing steered away
    from a discontinuity (ReLU, hard-sigmoid and MAE kinks get their per-entry allowance from the oracle), against the fp64 oracle.
    B = 4: the oracle's cost."""
    import bench
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, kernel_tags, oracle_reference
    _no_force_overrides()
    B = 4
    model = _cfg4_model(seed=7)
    w = model.get_weights()
    # (zero biases, except the ConvLSTM cells' unit forget bias: Keras' default, blocks.py:350-355)
    assert all(set(np.unique(v)) <= {0.0, 1.0} for k, v in w.items() if k.endswith('bias'))
    x, aux, y = bench.synthetic_batch_cfg4(1004, B)
    out = model([x, aux])
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([x, aux], y))
    for must in ('convlstm_seq_fwd<5,8>', 'convlstm_seq_bwd<5,8>', 'convlstm_seq_bwd<3,8>'):
        assert must in tags, (must, sorted(tags))
    ref = oracle_reference('supervised', 'recnet_postupsampling', CFG4_OCFG, w, x, aux, y, loss='mae', workers=ORACLE_WORKERS)
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    _slack_is_small(assert_matches_reference(g_hip, ref, what='cfg4 on the bench workload (box-blurred frames, zero biases, B = 4)', full=True),
                    limit=0.03)


def _cfg5_pair():
    import dl4ds_amd.models as PM
    H = 512
    gen = PM.unet_pin('unet', 5, 1, hr_size=(H, H), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=3)
    disc = PM.residual_discriminator(5, 'pin', False, 8, (H // 8, H // 8), n_filters=8, hr_size=(H, H), seed=4)
    assert gen.count_params() == 13566325 and disc.count_params() == 16177
    return gen, disc


CFG5_GCFG = dict(n_filters=8, n_blocks=6, decoder_upsampling='dc')
CFG5_DCFG = dict(upsampling='pin', scale=8, n_filters=8, n_res_blocks=4, lr_size=(64, 64))


def test_cfg5_generator_full_size_against_the_oracle():
    """BASELINE configs[4]'s generator (U-Net, 9x9 transposed-convolution decoder, 13.6 M parameters, 512^2) as a
    supervised MAE step against the fp64 oracle.  sp_preups.py:230-315."""
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, kernel_tags, oracle_reference
    _no_force_overrides()
    B = 8
    gen, _ = _cfg5_pair()
    w = _randomise_biases(gen)
    rng = np.random.default_rng(1005)
    lr = rng.standard_normal((B, 512, 512, 5)).astype(np.float32)
    st = rng.standard_normal((B, 512, 512, 1)).astype(np.float32)
    out = gen([lr, st])
    y = _targets_clear_of_the_kink(out, rng)
    eng = SupervisedEngine(gen, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([lr, st], y))
    for must in ('conv_narrow_pair_ws<4>', 'conv_narrow16_ws<4>', 'conv_narrow_wgrad<8>'):
        assert must in tags, (must, sorted(tags))
    assert any(t.startswith('conv_stream_ws<5,') for t in tags), sorted(tags)
    ref = oracle_reference('supervised', 'unet_pin', CFG5_GCFG, w, lr, st, y, loss='mae', workers=ORACLE_WORKERS)
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    _slack_is_small(assert_matches_reference(g_hip, ref, what='cfg5 generator full size', full=True))


def test_cfg5_generator_on_the_bench_workload_itself_against_the_oracle():
    """the generator of `bench.py --config cfg5` on the bench's own arrays (bench.synthetic_batch_cfg5(1005, B): five
    64^2 block-mean fields re-expanded x 8, a blurred static field, the blurred HR target) and its own zero-bias weights (seed 7), as
    a supervised MAE step against the fp64 oracle; nothing steered.  B = 4."""
    import bench
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, kernel_tags, oracle_reference
    _no_force_overrides()
    B = 4
    gen = PM.unet_pin('unet', 5, 1, hr_size=(512, 512), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=7)
    w = gen.get_weights()
    assert all(np.abs(v).max() == 0.0 for k, v in w.items() if k.endswith('bias'))
    x, aux, y = bench.synthetic_batch_cfg5(1005, B)
    out = gen([x, aux])
    eng = SupervisedEngine(gen, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([x, aux], y))
    for must in ('conv_narrow_pair_ws<4>', 'conv_narrow16_ws<4>', 'conv_narrow_wgrad<8>'):
        assert must in tags, (must, sorted(tags))
    ref = oracle_reference('supervised', 'unet_pin', CFG5_GCFG, w, x, aux, y, loss='mae', workers=ORACLE_WORKERS)
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    _slack_is_small(assert_matches_reference(g_hip, ref, what='cfg5 generator on the bench workload (re-expanded block means, zero biases, B = 4)',
                                             full=True), limit=0.03)


def test_cfg5_full_size_cgan_step_against_the_oracle():
    """The WHOLE configs[4] CGAN step at 512^2 through dl4ds_cgan_step with an injected dropout mask against the fp64
    restatement of train_step (cgan.py:575-639): four losses, both gradient sets per tensor."""
    from dl4ds_amd.training import CGANEngine
    from tests.parity import assert_matches_reference, oracle_reference
    _no_force_overrides()
    B = 4
    gen, disc = _cfg5_pair()
    rng = np.random.default_rng(1005)
    gw = _randomise_biases(gen, seed=int(rng.integers(1 << 30)))
    dw = _randomise_biases(disc, seed=int(rng.integers(1 << 30)))
    lr = rng.random((B, 512, 512, 5)).astype(np.float32)
    st = rng.random((B, 512, 512, 1)).astype(np.float32)
    hr = _targets_clear_of_the_kink(gen([lr, st]), rng)              # (the pixel loss compares the generated field with it)
    mask = (rng.random((2 * B, 16)) > 0.4).astype(np.float32)
    eng = CGANEngine(gen, disc, loss='mae', learning_rate=2e-4, beta_1=0.5)
    out = eng.step([lr, st], hr, dropout_keep=mask, apply_update=False)
    gg, gd = gen.get_gradients(), disc.get_gradients()
    ref = oracle_reference('cgan', 'unet_pin', CFG5_GCFG, gw, lr, st, hr, loss='mae', dcfg=CFG5_DCFG, dweights=dw, mask=mask,
                           workers=B)
    for i, k in enumerate(('gen_total', 'gen_gan', 'gen_px', 'disc')):
        assert out[i] == pytest.approx(ref['losses'][i], rel=1e-4), k
    # (B = 4: the bias of the merge block is a sum of four samples' terms, noise floor 1.6 % of it)
    _slack_is_small(assert_matches_reference(gd, ref, 'gradsD', what='cfg5 CGAN step full size: discriminator', full=True), limit=0.03)
    _slack_is_small(assert_matches_reference(gg, ref, 'gradsG', what='cfg5 CGAN step full size: generator (adversarial + 100 x MAE)', full=True))


def test_cfg2_after_training_steps_on_the_bench_workload_against_the_oracle():
    """The state `bench.py` actually times is not the initial one: after 25 Adam steps on the bench batch the biases are no longer zero,
    the residuals have shrunk and the ReLU patterns have moved.  The gradients AT THAT STATE (weights read back from the device) against
    the fp64 oracle, nothing steered; B = 16."""
    import bench
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import assert_matches_reference, oracle_reference
    _no_force_overrides()
    B = 16
    model = _cfg2(seed=7)
    x, y = bench.synthetic_batch(1002, B)
    eng = SupervisedEngine(model, loss='mae', learning_rate=(1e-3, 1e-4), lr_decay_after=1e5)
    losses = [eng.step([x], y) for _ in range(25)]
    assert losses[-1] < 0.5 * losses[0], losses                          # (it trains: 0.50 -> below 0.1 on the blurred fields)
    w = model.get_weights()
    assert any(np.abs(v).max() > 0.0 for k, v in w.items() if k.endswith('bias'))
    out = model([x])
    l_hip, g_hip = eng.loss_and_grads([x], y)
    ref = oracle_reference('supervised', 'net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4), w, x, None, y,
                           loss='mae', workers=ORACLE_WORKERS)
    _fwd_close(out, ref['pred'])
    assert l_hip == pytest.approx(ref['loss'], rel=1e-4)
    _slack_is_small(assert_matches_reference(g_hip, ref, what='cfg2 on the bench workload after 25 Adam steps (B = 16)', full=True), limit=0.05)


def test_cfg5_cgan_step_on_the_bench_workload_itself_against_the_oracle():
    """The CGAN step `bench.py --config cfg5` times -- its arrays (bench.synthetic_batch_cfg5), its zero-bias weights (generator seed 7,
    discriminator seed 8), an injected dropout mask -- against the fp64 restatement of train_step (cgan.py:575-639); nothing steered.
    B = 2: the oracle's cost."""
    import bench
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    from tests.parity import assert_matches_reference, oracle_reference
    _no_force_overrides()
    B, H = 2, 512
    gen = PM.unet_pin('unet', 5, 1, hr_size=(H, H), n_filters=8, n_blocks=6, decoder_upsampling='dc', seed=7)
    disc = PM.residual_discriminator(5, 'pin', False, 8, (H // 8, H // 8), n_filters=8, hr_size=(H, H), seed=8)
    gw, dw = gen.get_weights(), disc.get_weights()
    x, aux, y = bench.synthetic_batch_cfg5(1005, B)
    mask = (np.random.default_rng(5).random((2 * B, 16)) > 0.4).astype(np.float32)
    eng = CGANEngine(gen, disc, loss='mae', learning_rate=2e-4, beta_1=0.5)
    out = eng.step([x, aux], y, dropout_keep=mask, apply_update=False)
    gg, gd = gen.get_gradients(), disc.get_gradients()
    ref = oracle_reference('cgan', 'unet_pin', CFG5_GCFG, gw, x, aux, y, loss='mae', dcfg=CFG5_DCFG, dweights=dw, mask=mask, workers=B)
    for i, k in enumerate(('gen_total', 'gen_gan', 'gen_px', 'disc')):
        assert out[i] == pytest.approx(ref['losses'][i], rel=1e-4), k
    _slack_is_small(assert_matches_reference(gd, ref, 'gradsD', what='cfg5 CGAN step on the bench workload: discriminator (B = 2)', full=True), limit=0.05)
    _slack_is_small(assert_matches_reference(gg, ref, 'gradsG', what='cfg5 CGAN step on the bench workload: generator (B = 2)', full=True), limit=0.05)


def test_cfg2_default_dispatch_and_fp32_pipe_follow_one_loss_trajectory(monkeypatch):
    """Round 6: the product's default dispatch puts the eight single-pass <= 48-channel 3x3 layers of the bench step (B = 64) on
    conv_split_kernel (fp32 products as six bf16 MFMA terms); DL4DS_NO_SPLIT=1 keeps them on the fp32 pipe.  Twelve Adam steps on the
    bench batch from the same initial weights, once each way: the two loss curves agree to 2e-5 of their value (what two fp32
    summation orders differ by), and the kernel tags show that the two runs really took the two paths."""
    import bench
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import kernel_tags
    _no_force_overrides()
    B = 64
    x, y = bench.synthetic_batch(1002, B)
    curves, tags = {}, {}
    for name in ('default', 'fp32 pipe'):
        monkeypatch.delenv('DL4DS_SPLIT', raising=False)
        if name == 'default':
            monkeypatch.delenv('DL4DS_NO_SPLIT', raising=False)
        else:
            monkeypatch.setenv('DL4DS_NO_SPLIT', '1')
        model = _cfg2(seed=7)
        eng = SupervisedEngine(model, loss='mae', learning_rate=(1e-3, 1e-4), lr_decay_after=1e5)
        first, tags[name] = kernel_tags(lambda: eng.step([x], y))
        curves[name] = [first] + [eng.step([x], y) for _ in range(11)]
        del eng, model
    assert tags['default'].get('conv_split<3,3>') == 8 and 'conv_split<3,3>' not in tags['fp32 pipe'], (tags['default'], tags['fp32 pipe'])
    a, b = np.asarray(curves['default']), np.asarray(curves['fp32 pipe'])
    assert a[-1] < 0.7 * a[0], a
    assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max(), (a, b)

def test_split_filter_fragments_follow_the_weights():
    """conv_split_kernel's filter fragments (the weights as bf16 triples in MFMA fragment order) are built by one launch per pass for all the
    layers of a graph ('split_filters', from the second pass on; conv_split.hip) and must never outlive the weights they were made from:
    after set_weights a model gives the bits a fresh model with those weights gives.  cfg2 at B = 8: the eight single-pass <= 48-channel
    layers on the six-term kernel (16-row strips), 4 of them in the forward pass."""
    from tests.parity import kernel_tags
    _no_force_overrides()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((8, 128, 128, 1)).astype(np.float32)
    a = _cfg2(seed=21)
    y0, t0 = kernel_tags(lambda: a([x]))
    assert t0.get('conv_split<3,3>') == 4 and 'split_filters' not in t0, sorted(t0)          # first pass: every layer builds its own
    y1, t1 = kernel_tags(lambda: a([x]))
    assert t1.get('conv_split<3,3>') == 4 and t1.get('split_filters') == 1, sorted(t1)        # then one launch for all of them
    assert np.array_equal(y0, y1)
    w2 = {k: (v * np.float32(0.75) + np.float32(0.01)).astype(np.float32) for k, v in a.get_weights().items()}
    a.set_weights(w2)
    y2 = a([x])
    b = _cfg2(seed=33)
    b.set_weights(w2)
    assert np.array_equal(y2, b([x]))
    assert not np.array_equal(y2, y1)
