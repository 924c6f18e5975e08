"""Whole-model parity on MI355X: product builders + HIP graph runtime vs the oracle restatement
(independent builders, torch-CPU fp64 autograd) on identical seeded weights and inputs.
Tolerance 1e-3 relative (north_star); observed ~1e-5."""
import numpy as np
import pytest
import torch

from oracle import np_ops as N
from oracle import torch_ops as T
from oracle import models as M
from oracle import train as TR
from tests.parity import (BAND, assert_grads_close, assert_matches_reference, banded_reference, first_quiet, oracle_reference,
                          targets_clear_of_the_kink)

pytestmark = pytest.mark.gpu


def rel(a, ref):
    ref = np.asarray(ref, np.float64)
    return np.abs(np.asarray(a, np.float64) - ref).max() / max(np.abs(ref).max(), 1e-12)


def build_pair(kind, cfg, x_shape, s_shape, seed=11, ctx_kw=None):
    """Product model + oracle params holding the same weights.  ``ctx_kw``: the normalization / dropout builder
    arguments, which the oracle takes through its ``Ctx``."""
    import dl4ds_amd.models as PM
    ocfg_in = dict(cfg)
    cfg = dict(cfg, **(ctx_kw or {}))
    if kind == 'net_postupsampling':
        model = PM.net_postupsampling(n_channels=x_shape[-1], n_aux_channels=0 if s_shape is None else s_shape[-1],
                                      lr_size=x_shape[1:3], seed=seed, **cfg)
    elif kind == 'net_pin':
        model = PM.net_pin(n_channels=x_shape[-1], n_aux_channels=0 if s_shape is None else s_shape[-1],
                           hr_size=x_shape[1:3], seed=seed, **cfg)
    elif kind == 'unet_pin':
        model = PM.unet_pin('unet', n_channels=x_shape[-1], n_aux_channels=0 if s_shape is None else s_shape[-1],
                            hr_size=x_shape[1:3], seed=seed, **cfg)
    elif kind == 'recnet_postupsampling':
        model = PM.recnet_postupsampling(n_channels=x_shape[-1], n_aux_channels=0 if s_shape is None else s_shape[-1],
                                         lr_size=x_shape[2:4], seed=seed, **cfg)
    elif kind == 'recnet_pin':
        model = PM.recnet_pin(n_channels=x_shape[-1], n_aux_channels=0 if s_shape is None else s_shape[-1],
                              hr_size=x_shape[2:4], seed=seed, **cfg)
    rng = np.random.default_rng(seed + 1)
    w = model.get_weights()
    for k in w:                                   # make biases non-zero so they are exercised
        if k.endswith('moving_variance'):
            w[k] = (0.5 + rng.random(w[k].shape)).astype(np.float32)
        elif k.endswith('gamma'):
            w[k] = (1 + rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
        elif w[k].ndim == 1 or k.endswith('bias'):
            w[k] = (rng.standard_normal(w[k].shape) * 0.05).astype(np.float32)
    model.set_weights(w)
    ocfg = dict(ocfg_in)
    if kind == 'unet_pin':
        ocfg.pop('backbone_block', None)
    extra = dict(ctx=M.Ctx(**ctx_kw)) if ctx_kw else {}
    P0 = M.init_params(kind, (1,) + tuple(x_shape[1:]), None if s_shape is None else (1,) + tuple(s_shape[1:]),
                       dtype=np.float64, **ocfg, **extra)
    assert set(P0.keys()) == set(w.keys()), (sorted(set(P0) ^ set(w)))
    P = M.Params()
    for k in P0:
        assert tuple(P0[k].shape) == w[k].shape, k
        P[k] = w[k].astype(np.float64)
    return model, P, ocfg


SUP_CASES = [
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4), (2, 16, 16, 1), None),
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, n_blocks=2, attention=True),
     (2, 12, 20, 2), (2, 24, 40, 1)),
    ('net_postupsampling', dict(backbone_block='densenet', upsampling='rc', scale=2, n_blocks=2, localcon_layer=True),
     (2, 10, 14, 1), (2, 20, 28, 2)),
    ('net_postupsampling', dict(backbone_block='convnet', upsampling='spc', scale=5, n_blocks=1, activation='elu',
                                output_activation='sigmoid'), (1, 8, 8, 1), None),
    ('net_pin', dict(backbone_block='resnet'), (2, 32, 32, 2), None),
    ('unet_pin', dict(n_filters=4, n_blocks=3, decoder_upsampling='dc'), (2, 32, 32, 3), (2, 32, 32, 1)),
    ('unet_pin', dict(n_filters=4, n_blocks=2, decoder_upsampling='spc', attention=True), (1, 16, 24, 2), None),
    ('unet_pin', dict(n_filters=8, n_blocks=2, decoder_upsampling='rc'), (1, 16, 16, 5), (1, 16, 16, 1)),
    # Resizing(interpolation='nearest') in the resize-convolution upsamplers (blocks.py:473-489)
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='rc', scale=3, n_blocks=1, n_filters=4,
                                rc_interpolation='nearest'), (2, 7, 9, 2), (2, 21, 27, 1)),
    ('unet_pin', dict(n_filters=4, n_blocks=2, decoder_upsampling='rc', rc_interpolation='nearest'), (1, 16, 20, 2), None),
    # Resizing(interpolation='bicubic')
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='rc', scale=2, n_blocks=1, n_filters=4,
                                rc_interpolation='bicubic'), (2, 9, 7, 2), (2, 18, 14, 1)),
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='rc', scale=4, n_blocks=1, n_filters=4,
                                rc_interpolation='lanczos3'), (2, 6, 5, 2), (2, 24, 20, 1)),
    ('unet_pin', dict(n_filters=4, n_blocks=2, decoder_upsampling='rc', rc_interpolation='mitchellcubic'), (1, 16, 20, 2), None),
    # odd grids: MaxPooling2D drops a row / column, PadConcat zero-pads the decoder side back (blocks.py:629-656)
    ('unet_pin', dict(n_filters=4, n_blocks=2, decoder_upsampling='rc'), (2, 25, 30, 2), None),
    ('unet_pin', dict(n_filters=4, n_blocks=3, decoder_upsampling='spc'), (1, 37, 23, 1), (1, 37, 23, 1)),
    ('recnet_postupsampling', dict(backbone_block='densenet', upsampling='rc', scale=2, time_window=3, n_filters=4,
                                   n_blocks=1, attention=True, localcon_layer=True), (2, 3, 8, 8, 1), (2, 16, 16, 1)),
    ('recnet_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, time_window=2, n_filters=4,
                                   n_blocks=1), (1, 2, 10, 6, 2), None),
    ('recnet_pin', dict(backbone_block='convnet', time_window=4, n_filters=4, n_blocks=1), (1, 4, 8, 8, 2), None),
]


def tie_free_inputs(kind, P, ocfg, xs, ss, first_seed=5):
    """Seeded inputs for which no ReLU input of the fp64 oracle lies within 1.25 BAND = 2.5e-6 (relative to the largest one of its
    layer: the oracle's own normalisation of the displacement, oracle/torch_ops.py) of zero.  Such a value (seen: |pre| = 5e-8 at
    scale 0.7 for seed 5 on net_pin) takes either branch depending on fp32 summation order, so the comparison would test the tie
    and not the kernels; with none inside the band the oracle's two displaced evaluations agree and NOTHING is granted to any
    entry (round 4 searched at 3e-7 only: units between that and the band left 0.8 % of net_pin's entries on slack).
    Returns (rng, x, s, oracle forward)."""
    orig = N.relu
    best = None
    for seed in range(first_seed, first_seed + 12):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal(xs).astype(np.float32)
        s = None if ss is None else rng.standard_normal(ss).astype(np.float32)
        margins = []

        def spy(v):
            a = np.abs(v)
            margins.append(a.min() / max(a.max(), 1e-30))
            return orig(v)
        N.relu = spy
        try:
            ref = M.MODELS[kind](N, P, x.astype(np.float64), None if s is None else s.astype(np.float64), **ocfg)
        finally:
            N.relu = orig
        m = min(margins) if margins else 1.0
        if best is None or m > best[0]:
            best = (m, seed, x, s, ref)
        if m > 1.25 * BAND:
            break
    # (models with ~10^6 ReLU units have no seed without a unit inside the band: the seed whose closest unit is farthest is
    #  taken, and the few entries that unit feeds are what the per-element band is for -- the caps of tests/parity.py apply)
    _, seed, x, s, ref = best
    rng = np.random.default_rng(seed)
    rng.standard_normal(xs)
    if ss is not None:
        rng.standard_normal(ss)
    return rng, x, s, ref


@pytest.mark.parametrize('kind,cfg,xs,ss', SUP_CASES)
@pytest.mark.parametrize('loss', ['mae', 'mse'])
def test_supervised_forward_grads_and_adam_step(kind, cfg, xs, ss, loss):
    from dl4ds_amd.training import SupervisedEngine
    model, P, ocfg = build_pair(kind, cfg, xs, ss)
    rng, x, s, ref = tie_free_inputs(kind, P, ocfg, xs, ss)
    inputs = [x] if s is None else [x, s]
    # forward
    out = model(inputs)
    assert out.shape == ref.shape
    assert rel(out, ref) < 1e-3
    # loss + grads
    y = targets_clear_of_the_kink(ref, rng)          # (gradients that are sums, not random walks: oracle/reference.py)
    PT = M.convert(P, T, requires_grad=True)
    opt = TR.Adam(PT, lr=1e-3)
    lv, grads, _ = TR.supervised_step(kind, ocfg, PT, T.asarray(x.astype(np.float64)),
                                      None if s is None else T.asarray(s.astype(np.float64)),
                                      T.asarray(y.astype(np.float64)), loss=loss, opt=None)
    eng = SupervisedEngine(model, loss=loss, learning_rate=1e-3)
    l_hip, g_hip = eng.loss_and_grads(inputs, y)
    assert l_hip == pytest.approx(lv, rel=1e-4)
    # every tensor at its own scale, with the oracle's discontinuity band and fp32 noise floor (tests/parity.py)
    ref = oracle_reference('supervised', kind, ocfg, model.get_weights(), x, s, y, loss=loss)
    assert_matches_reference(g_hip, ref, what=(kind, loss))
    steady = {}
    for k in grads:
        gr = grads[k].numpy()
        # Elements whose gradient already differs by > 1 % of its own size: a pre-activation within fp32 rounding of 0
        # takes the other ReLU branch in the fp64 oracle (seen: |pre| = 5e-8 at scale 0.7), which moves a handful of
        # small gradient entries.  Adam divides by |g|, so those entries may move by a whole step; they are excluded
        # from the update comparison below (and must stay a small minority).
        steady[k] = np.abs(g_hip[k] - gr) <= 0.01 * (np.abs(gr) + 1e-3 * np.abs(gr).max())
    n_all = sum(m.size for m in steady.values())
    n_unsteady = sum(int((~m).sum()) for m in steady.values())
    assert n_unsteady <= 0.02 * n_all, (n_unsteady, n_all)
    # three optimiser steps
    w0 = model.get_weights()
    for it in range(3):
        lv, grads, _ = TR.supervised_step(kind, ocfg, PT, T.asarray(x.astype(np.float64)),
                                          None if s is None else T.asarray(s.astype(np.float64)),
                                          T.asarray(y.astype(np.float64)), loss=loss, opt=opt)
        l_hip = eng.step(inputs, y)
        assert l_hip == pytest.approx(lv, rel=2e-3), it
        if it == 0:
            # The first update must be Keras-Adam applied to the gradients the HIP path itself reported, to fp32 rounding
            # (2e-6 = 0.2 % of one step) for EVERY entry: m = (1-b1) g, v = (1-b2) g^2, lr_t = lr sqrt(1-b2)/(1-b1).
            # (The comparison with the oracle's trajectory below has to be loose: Adam divides by |g|, so an entry whose
            # gradient differs in the last bits of a near-zero value may legitimately move by a whole step.)
            w1 = model.get_weights()
            b1, b2, eps, lr = 0.9, 0.999, 1e-7, 1e-3
            for kk in w1:
                g64 = g_hip[kk].astype(np.float64)
                m1, v1 = (1 - b1) * g64, (1 - b2) * g64 * g64
                ref1 = w0[kk].astype(np.float64) - lr * np.sqrt(1 - b2) / (1 - b1) * m1 / (np.sqrt(v1) + eps)
                assert np.abs(w1[kk] - ref1).max() < 2e-6, kk
    w = model.get_weights()
    for k in w:
        upd_ref = PT[k].detach().numpy() - w0[k]
        upd = w[k] - w0[k]
        # Adam's first steps are ~lr*sign(g): compare the updates at 3 steps * lr scale
        dev = np.abs(upd - upd_ref)
        assert dev[steady[k]].max(initial=0.0) < 0.15 * 3e-3 + 1e-6, k
        assert dev.max() < 3.3e-3, k              # never more than the three steps themselves
    m, v, step = eng.optimizer_state()
    assert step == 3


def test_headline_model_through_producer_consumer_kernels(monkeypatch):
    """The headline configuration's layers take conv_stream_ws_kernel / conv_narrow_pair_ws_kernel only on grids large
    enough to give every workgroup two items (bench sizes); DL4DS_STREAM_FORCE_WS makes this reduced model take them too:
    forward, loss, all gradients against the oracle."""
    monkeypatch.setenv('DL4DS_STREAM_FORCE_WS', '2')
    from dl4ds_amd.training import SupervisedEngine
    kind, cfg, xs, ss = SUP_CASES[0]
    model, P, ocfg = build_pair(kind, cfg, xs, ss)
    rng, x, s, ref = tie_free_inputs(kind, P, ocfg, xs, ss)
    inputs = [x] if s is None else [x, s]
    out = model(inputs)
    assert rel(out, ref) < 1e-3
    y = targets_clear_of_the_kink(ref, rng)          # (gradients that are sums, not random walks: oracle/reference.py)
    PT = M.convert(P, T, requires_grad=True)
    lv, grads, _ = TR.supervised_step(kind, ocfg, PT, T.asarray(x.astype(np.float64)),
                                      None if s is None else T.asarray(s.astype(np.float64)),
                                      T.asarray(y.astype(np.float64)), loss='mae', opt=None)
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    l_hip, g_hip = eng.loss_and_grads(inputs, y)
    assert l_hip == pytest.approx(lv, rel=1e-4)
    ref = oracle_reference('supervised', kind, ocfg, model.get_weights(), x, s, y, loss='mae')
    assert_matches_reference(g_hip, ref, what='producer / consumer kernels')


@pytest.mark.parametrize('sx', ['all', '1'])
def test_wide_model_through_the_winograd_kernels(monkeypatch, sx):
    """A residual net with 16 .. 64 filters takes the Winograd kernels (conv_wino.hip, conv_wino_wgrad.hip) only on bench-sized
    grids; DL4DS_WINO_FORCE makes this 32 x 32 one take them: layers of 32, 48 and 64 channels, i.e. one and two passes over
    the input channels, residual adds and ReLU masks in the last pass of two (epilogue forms that the headline model does not
    reach), weight gradients with two input-channel chunks.  Forward, loss and every gradient against the oracle."""
    monkeypatch.setenv('DL4DS_WINO_FORCE', sx)
    from dl4ds_amd.training import SupervisedEngine
    from tests.parity import kernel_tags
    kind, cfg, xs, ss = 'net_pin', dict(backbone_block='resnet', n_filters=16, n_blocks=4), (2, 32, 32, 2), None
    model, P, ocfg = build_pair(kind, cfg, xs, ss)
    rng, x, s, ref = tie_free_inputs(kind, P, ocfg, xs, ss)
    out, tags_f = kernel_tags(lambda: model([x]))
    assert any(t.startswith('conv_wino<') for t in tags_f), sorted(tags_f)
    assert rel(out, ref) < 1e-3
    y = targets_clear_of_the_kink(ref, rng)          # (gradients that are sums, not random walks: oracle/reference.py)
    PT = M.convert(P, T, requires_grad=True)
    lv, grads, _ = TR.supervised_step(kind, ocfg, PT, T.asarray(x.astype(np.float64)), None, T.asarray(y.astype(np.float64)),
                                      loss='mae', opt=None)
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    (l_hip, g_hip), tags = kernel_tags(lambda: eng.loss_and_grads([x], y))
    assert any(t.startswith('conv_wino_wgrad<') for t in tags), sorted(tags)
    assert l_hip == pytest.approx(lv, rel=1e-4)
    ref = oracle_reference('supervised', kind, ocfg, model.get_weights(), x, s, y, loss='mae')
    assert_matches_reference(g_hip, ref, what='Winograd kernels')


@pytest.mark.parametrize('n_aux', [0, 2])
@pytest.mark.parametrize('ups,scale', [('spc', 4), ('spc', 2), ('rc', 2), ('spc', 5)])
def test_folded_upsampling_tail_equals_unfolded(monkeypatch, ups, scale, n_aux):
    """The last convolution of the upsampling block composed with TransitionLast (csrc/graph_ops3.hip, FoldedConvOp)
    against the same model built with DL4DS_NO_FOLD=1 (two separate layers, the reference's evaluation order): same
    variables, and forward output, loss, every gradient (conv2x is shared with the first x2 stage, so its gradient is
    accumulated from both) and the weights after three Adam steps agree to fp32 rounding.  ``n_aux`` > 0: the model has
    HR auxiliary channels, TransitionLast reads Concatenate([x, ConvBlock_aux(s)]) and the composed layer takes the
    auxiliary part as a separate 1x1 convolution added in its epilogue (FoldedConvOp's auxiliary form)."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    cfg = dict(backbone_block='resnet', upsampling=ups, scale=scale, n_channels=2, n_aux_channels=n_aux, lr_size=(12, 10),
               n_blocks=2, n_filters=8, seed=5)
    monkeypatch.delenv('DL4DS_NO_FOLD', raising=False)
    folded = PM.net_postupsampling(**cfg)
    monkeypatch.setenv('DL4DS_NO_FOLD', '1')
    plain = PM.net_postupsampling(**cfg)
    monkeypatch.delenv('DL4DS_NO_FOLD')
    kinds = [k for k, _, _ in folded.graph.layers]
    assert any('folded' in k for k in kinds) and not any('folded' in k for k, _, _ in plain.graph.layers)
    assert len(plain.graph.layers) == len(folded.graph.layers) + (1 if n_aux == 0 else 2)      # (+ the Concatenate)
    rng = np.random.default_rng(3)
    w = plain.get_weights()
    assert sorted(w) == sorted(folded.get_weights())          # (same variables; the auxiliary block is created earlier)
    for k in w:
        if k.endswith('bias'):
            w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    plain.set_weights(w); folded.set_weights(w)
    x = rng.standard_normal((3, 12, 10, 2)).astype(np.float32)
    y = rng.standard_normal((3, 12 * scale, 10 * scale, 1)).astype(np.float32)
    ins = [x] if n_aux == 0 else [x, rng.standard_normal((3, 12 * scale, 10 * scale, n_aux)).astype(np.float32)]
    x = ins if n_aux else x
    assert rel(folded(x), plain(x)) < 2e-5
    e1, e2 = SupervisedEngine(folded, loss='mse', learning_rate=1e-3), SupervisedEngine(plain, loss='mse', learning_rate=1e-3)
    l1, g1 = e1.loss_and_grads(ins, y)
    l2, g2 = e2.loss_and_grads(ins, y)
    assert l1 == pytest.approx(l2, rel=1e-5)
    assert_grads_close(g1, g2, tol=1e-4, what='folded vs unfolded')
    for _ in range(3):
        a, b = e1.step(ins, y), e2.step(ins, y)
        assert a == pytest.approx(b, rel=1e-4)
    w1, w2 = folded.get_weights(), plain.get_weights()
    for k in w1:
        # Adam's first steps move every weight by ~lr regardless of the gradient's size: compare at that scale
        assert np.abs(w1[k] - w2[k]).max() < 0.1 * 3e-3, k


REC_TAIL_CASES = [
    # builder, its arguments, the LR / HR grid.  HR grids of 16 x 16 = 256 points take the LDS-staged kernels (weight gradients
    # on the matrix pipe inside the backward pass), 12 x 10 the plain ones (dz materialised, three 1x1 weight-gradient calls)
    ('recnet_postupsampling', dict(backbone_block='densenet', upsampling='rc', scale=2, n_blocks=1), (8, 8)),      # cfg4's shape
    ('recnet_postupsampling', dict(backbone_block='densenet', upsampling='rc', scale=2, n_blocks=1), (6, 5)),
    ('recnet_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, n_blocks=1), (8, 8)),
    ('recnet_postupsampling', dict(backbone_block='convnet', upsampling='rc', scale=2, n_blocks=1), (6, 5)),
    ('recnet_pin', dict(backbone_block='densenet', n_blocks=1), (16, 16)),
    ('recnet_pin', dict(backbone_block='densenet', n_blocks=1), (12, 10)),
    ('recnet_pin', dict(backbone_block='convnet', n_blocks=1), (16, 16)),
    ('recnet_pin', dict(backbone_block='resnet', n_blocks=1), (12, 10)),
]


@pytest.mark.parametrize('kind,kw,grid', REC_TAIL_CASES)
def test_recurrent_tail_as_one_op_equals_the_separate_layers(monkeypatch, kind, kw, grid):
    """[x, repeat(ConvBlock_aux(s)), LocalizedConvBlock(..)] -> TransitionLast as ONE op per direction (csrc/graph_ops4.hip,
    RecTailOp: spt_postups.py:133-151 / spt_preups.py:114-132) against the same model built with DL4DS_NO_REC_TAIL_FUSION=1
    (repeat_time, two Concatenates, 1x1 conv, LocallyConnected2D, 1x1 conv): same variables; forward output, loss, every
    gradient and the weights after three Adam steps agree to fp32 rounding.  Covers every channel combination that is built
    (REC_TAIL_SHAPES) in both kernel forms, and two batch sizes on one graph (the ConvLSTM tile flags are never reset)."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    T_, B = 3, 2
    h, w = grid
    if kind == 'recnet_postupsampling':
        cfg = dict(n_channels=2, n_aux_channels=2, lr_size=(h, w), time_window=T_, n_filters=8, attention=True,
                   localcon_layer=True, seed=7, **kw)
        H, W = h * kw['scale'], w * kw['scale']
        build = PM.recnet_postupsampling
        xs = (B, T_, h, w, 2)
    else:
        cfg = dict(n_channels=2, n_aux_channels=2, hr_size=(h, w), time_window=T_, n_filters=8, localcon_layer=True, seed=7, **kw)
        H, W = h, w
        build = PM.recnet_pin
        xs = (B, T_, h, w, 2)
    monkeypatch.delenv('DL4DS_NO_REC_TAIL_FUSION', raising=False)
    fused = build(**cfg)
    monkeypatch.setenv('DL4DS_NO_REC_TAIL_FUSION', '1')
    plain = build(**cfg)
    monkeypatch.delenv('DL4DS_NO_REC_TAIL_FUSION')
    assert any(k == 'rec_tail' for k, _, _ in fused.graph.layers), [k for k, _, _ in fused.graph.layers]
    assert not any(k == 'rec_tail' for k, _, _ in plain.graph.layers)
    rng = np.random.default_rng(11)
    wts = plain.get_weights()
    assert sorted(wts) == sorted(fused.get_weights())
    for k in wts:
        if k.endswith('bias'):
            wts[k] = (rng.standard_normal(wts[k].shape) * 0.1).astype(np.float32)
    plain.set_weights(wts); fused.set_weights(wts)
    x = rng.standard_normal(xs).astype(np.float32)
    s = rng.standard_normal((B, H, W, 2)).astype(np.float32)
    y = rng.standard_normal((B, T_, H, W, 1)).astype(np.float32)
    assert rel(fused([x, s]), plain([x, s])) < 2e-5
    e1, e2 = SupervisedEngine(fused, loss='mse', learning_rate=1e-3), SupervisedEngine(plain, loss='mse', learning_rate=1e-3)
    l1, g1 = e1.loss_and_grads([x, s], y)
    l2, g2 = e2.loss_and_grads([x, s], y)
    assert l1 == pytest.approx(l2, rel=1e-5)
    assert_grads_close(g1, g2, tol=1e-4, what='rec_tail vs separate layers')
    # a smaller batch on the same graph, then the first one again: identical results (nothing stale in the op's buffers)
    l1h, g1h = e1.loss_and_grads([x[:1], s[:1]], y[:1])
    l2h, g2h = e2.loss_and_grads([x[:1], s[:1]], y[:1])
    assert l1h == pytest.approx(l2h, rel=1e-5)
    assert_grads_close(g1h, g2h, tol=1e-4, what='rec_tail vs separate layers, B = 1')
    l1b, g1b = e1.loss_and_grads([x, s], y)
    assert l1b == l1
    for k in g1:
        assert np.array_equal(g1[k], g1b[k]), k
    for _ in range(3):
        a, b = e1.step([x, s], y), e2.step([x, s], y)
        assert a == pytest.approx(b, rel=1e-4)
    w1, w2 = fused.get_weights(), plain.get_weights()
    for k in w1:
        assert np.abs(w1[k] - w2[k]).max() < 0.1 * 3e-3, k


def test_cfg2_parameter_count_and_name():
    import dl4ds_amd.models as PM
    m = PM.net_postupsampling('resnet', 'spc', 4, 1, 0, (16, 16))
    assert m.count_params() == 204405
    assert m.name == 'resnet_spc'
    assert m.output_shape == (64, 64, 1)
    m = PM.net_pin('resnet', 2, 0, (16, 16))
    assert m.count_params() == 121341 and m.name == 'resnet_pin'


def _cgan_banded(gkind, gcfg, gw, dcfg, dw, lr, hr, st, mask):
    """tests/parity.py's reference of one whole-batch CGAN step: generator gradients under 'G:<name>', discriminator
    gradients under 'D:<name>' (BatchNormalization moving statistics carry no gradient and are left out)."""
    B = lr.shape[0]

    def call(dt):
        PG, PD = M.Params(), M.Params()
        for k, v in gw.items():
            PG[k] = np.asarray(v, dt)
        for k, v in dw.items():
            PD[k] = np.asarray(v, dt)
        t = lambda a: T.asarray(a.astype(dt))
        r = TR.cgan_step(gkind, gcfg, M.convert(PG, T, requires_grad=True), dcfg, M.convert(PD, T, requires_grad=True), t(lr), t(hr),
                         t(st), dropout_masks=(t(mask[:B]), t(mask[B:])))
        g = {'G:' + k: v for k, v in r['gradsG'].items()}
        g.update({'D:' + k: v for k, v in r['gradsD'].items()})
        return [r['gen_total'], r['gen_gan'], r['gen_px'], r['disc']], g, None
    return banded_reference(call)


def test_cgan_step_matches_oracle():
    """cfg5-shaped (tiny): U-Net(dc decoder) generator + residual discriminator, one CGAN step with an injected
    dropout mask: losses, both gradient sets and the two Adam updates against the torch-CPU restatement."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    B, H = 2, 16
    gcfg = dict(n_filters=4, n_blocks=2, decoder_upsampling='dc')
    dcfg = dict(upsampling='pin', scale=8, n_filters=4, n_res_blocks=2)
    gen = PM.unet_pin('unet', 3, 1, hr_size=(H, H), seed=3, **gcfg)
    disc = PM.residual_discriminator(3, 'pin', False, 8, (H // 8, H // 8), n_filters=4, n_res_blocks=2, hr_size=(H, H),
                                     seed=4)
    rng = np.random.default_rng(9)
    for m in (gen, disc):
        w = m.get_weights()
        for k in w:
            if k.endswith('bias'):
                w[k] = (rng.standard_normal(w[k].shape) * 0.05).astype(np.float32)
        m.set_weights(w)
    PG = M.Params(); PD = M.Params()
    for k, v in gen.get_weights().items():
        PG[k] = v.astype(np.float64)
    for k, v in disc.get_weights().items():
        PD[k] = v.astype(np.float64)
    g0, d0 = gen.get_weights(), disc.get_weights()

    def draw(seed):
        r = np.random.default_rng(seed)
        lr = r.random((B, H, H, 3)).astype(np.float32)
        st = r.random((B, H, H, 1)).astype(np.float32)
        hr = r.random((B, H, H, 1)).astype(np.float32)
        mask = (r.random((2 * B, 8)) > 0.4).astype(np.float32)
        return lr, st, hr, mask, _cgan_banded('unet_pin', gcfg, g0, dcfg, d0, lr, hr, st, mask)
    lr, st, hr, mask, bref = first_quiet(draw, 90)           # (inputs on which the oracle's reference is well defined)
    PGt, PDt = M.convert(PG, T, requires_grad=True), M.convert(PD, T, requires_grad=True)
    optG, optD = TR.Adam(PGt, lr=2e-4, beta1=0.5), TR.Adam(PDt, lr=2e-4, beta1=0.5)
    t64 = lambda a: T.asarray(a.astype(np.float64))
    ref = TR.cgan_step('unet_pin', gcfg, PGt, dcfg, PDt, t64(lr), t64(hr), t64(st),
                       dropout_masks=(t64(mask[:B]), t64(mask[B:])), optG=optG, optD=optD)
    eng = CGANEngine(gen, disc, loss='mae', learning_rate=2e-4, beta_1=0.5)
    out = eng.step([lr, st], hr, dropout_keep=mask)
    assert out[0] == pytest.approx(ref['gen_total'], rel=1e-4)
    assert out[1] == pytest.approx(ref['gen_gan'], rel=1e-4)
    assert out[2] == pytest.approx(ref['gen_px'], rel=1e-4)
    assert out[3] == pytest.approx(ref['disc'], rel=1e-4)
    gg, gd = gen.get_gradients(), disc.get_gradients()
    assert_matches_reference({'G:' + k: v for k, v in gg.items()} | {'D:' + k: v for k, v in gd.items()}, bref, what='cgan')
    # first Adam step = -lr * sign(g) (to eps): compare against the oracle's updated weights
    for m, P0, Pt in ((gen, g0, PGt), (disc, d0, PDt)):
        w = m.get_weights()
        for k in w:
            assert np.abs((w[k] - P0[k]) - (Pt[k].detach().numpy() - P0[k])).max() < 0.2 * 2e-4 + 1e-7, k


@pytest.mark.parametrize('switch', ['DL4DS_NO_SHARED_BRANCH', 'DL4DS_NO_CGAN_RATIO', 'DL4DS_NO_TWO_ADD_INPLACE'])
def test_cgan_shared_conditioning_branch_equals_two_evaluations(monkeypatch, switch):
    """The discriminator sees [real ; fake] with the SAME conditioning array in both halves; its conditioning branch is evaluated
    once (Graph::plan_shared, csrc/graph.hip), the hand-over tensor copied to the second half and its gradient halves summed
    before the branch's backward.  Against the same step with the branch evaluated on both halves (DL4DS_NO_SHARED_BRANCH=1:
    what the reference's two discriminator calls do, cgan.py:598-599): losses and every gradient of both models.
    DL4DS_NO_CGAN_RATIO: the generator's adversarial gradient by per-sample rescaling of the discriminator-loss pass
    (csrc/cgan.hip, cgan_ratio_kernel) against the second backward pass through the discriminator it replaces.
    DL4DS_NO_TWO_ADD_INPLACE (round 5): a branch input that feeds one convolution and two fused adds gets its gradient without a copy
    (ConvOp::plan_two_adds: its buffer is the first add's dZ, the second add's dZ rides on the convolution's dgrad store) against the
    copy + accumulate passes it replaces."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    B, H = 3, 24
    rng = np.random.default_rng(21)
    lr = rng.random((B, H, H, 3)).astype(np.float32)
    st = rng.random((B, H, H, 1)).astype(np.float32)
    hr = rng.random((B, H, H, 1)).astype(np.float32)
    mask = (rng.random((2 * B, 8)) > 0.4).astype(np.float32)
    res = []
    for off in (False, True):
        if off:
            monkeypatch.setenv(switch, '1')
        else:
            monkeypatch.delenv(switch, raising=False)
        gen = PM.unet_pin('unet', 3, 1, hr_size=(H, H), seed=3, n_filters=4, n_blocks=2, decoder_upsampling='dc')
        disc = PM.residual_discriminator(3, 'pin', False, 8, (H // 8, H // 8), n_filters=4, n_res_blocks=2, hr_size=(H, H), seed=4)
        eng = CGANEngine(gen, disc, loss='mae', learning_rate=2e-4, beta_1=0.5)
        out = eng.step([lr, st], hr, dropout_keep=mask)
        res.append((out, gen.get_gradients(), disc.get_gradients()))
    (o0, g0, d0), (o1, g1, d1) = res
    for a, b in zip(o0, o1):
        assert a == pytest.approx(b, rel=1e-5)
    for name, (x, y) in [('G', (g0, g1)), ('D', (d0, d1))]:
        for k in x:
            scale = max(np.abs(y[k]).max(), 1e-12)
            assert np.abs(x[k] - y[k]).max() <= 2e-5 * scale + 1e-9, (name, k)
    # the branch's weight gradients are sums over a different association of the two halves: close, not identical
    changed = g0 if switch == 'DL4DS_NO_CGAN_RATIO' else d0
    other = g1 if switch == 'DL4DS_NO_CGAN_RATIO' else d1
    assert any(not np.array_equal(changed[k], other[k]) for k in changed)     # (the switch did select another evaluation)


CGAN_CASES = [
    # generator kind, generator cfg, discriminator cfg (oracle), LR grid, scale, time window
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4, n_blocks=1, n_filters=4),
     dict(upsampling='spc', scale=4, n_filters=4, n_res_blocks=1), (6, 8), 4, None),
    ('net_postupsampling', dict(backbone_block='convnet', upsampling='rc', scale=5, n_blocks=1, n_filters=4),
     dict(upsampling='rc', scale=5, n_filters=4, n_res_blocks=1, normalization='ln'), (8, 8), 5, None),
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, n_blocks=1, n_filters=4),
     dict(upsampling='spc', scale=2, n_filters=4, n_res_blocks=2, attention=True), (8, 7), 2, None),
    ('recnet_pin', dict(backbone_block='convnet', time_window=2, n_filters=4, n_blocks=1),
     dict(upsampling='pin', scale=2, n_filters=4, n_res_blocks=1), (8, 8), 1, 2),
    ('recnet_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4, time_window=3, n_filters=4,
                                   n_blocks=1), dict(upsampling='spc', scale=4, n_filters=4, n_res_blocks=1), (4, 6), 4, 3),
    # BatchNormalization in the discriminator's residual blocks (discriminator.py:38,50,70): the real and the generated
    # batch are normalised separately (two calls in the reference, two statistics groups of one pass here)
    ('net_pin', dict(backbone_block='convnet', n_blocks=1, n_filters=4),
     dict(upsampling='pin', scale=2, n_filters=4, n_res_blocks=1, normalization='bn'), (8, 10), 1, None),
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4, n_blocks=1, n_filters=4),
     dict(upsampling='spc', scale=4, n_filters=4, n_res_blocks=2, normalization='bn', attention=True), (6, 8), 4, None),
]


@pytest.mark.parametrize('gkind,gcfg,dcfg,lr_hw,scale,tw', CGAN_CASES)
def test_cgan_step_discriminator_variants(gkind, gcfg, dcfg, lr_hw, scale, tw):
    """CGAN step with the discriminator forms of discriminator.py:31-33,52-63,73-74: post-upsampling generators (HR
    branch reduced by stride-2 convolutions -- 'same' for scale 4, 'valid' + crop for scale 5 -- or bilinear resizing),
    LayerNormalization in the residual blocks, and the spatio-temporal form (ConvLSTM + LN stem, 3-D pooling)."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    B = 2
    h, w = lr_hw
    H, W = h * scale, w * scale
    lead = () if tw is None else (tw,)
    n_ch = 2
    if gkind == 'net_postupsampling':
        gen = PM.net_postupsampling(n_channels=n_ch, n_aux_channels=1, lr_size=(h, w), seed=3, **gcfg)
    elif gkind == 'net_pin':
        gen = PM.net_pin(n_channels=n_ch, n_aux_channels=1, hr_size=(H, W), seed=3, **gcfg)
    elif gkind == 'recnet_pin':
        gen = PM.recnet_pin(n_channels=n_ch, n_aux_channels=1, hr_size=(H, W), seed=3, **gcfg)
    else:
        gen = PM.recnet_postupsampling(n_channels=n_ch, n_aux_channels=1, lr_size=(h, w), seed=3, **gcfg)
    dkw = {k: v for k, v in dcfg.items() if k not in ('upsampling', 'scale')}
    disc = PM.residual_discriminator(n_ch, dcfg['upsampling'], tw is not None, dcfg['scale'], (h, w), hr_size=(H, W),
                                     time_window=tw, seed=4, **dkw)
    rng = np.random.default_rng(9)
    for m in (gen, disc):
        wts = m.get_weights()
        for k in wts:
            if k.endswith('bias') or k.endswith('beta'):
                wts[k] = (rng.standard_normal(wts[k].shape) * 0.05).astype(np.float32)
            elif k.endswith('gamma'):
                wts[k] = (1 + 0.1 * rng.standard_normal(wts[k].shape)).astype(np.float32)
            elif k.endswith('moving_variance'):
                wts[k] = (0.5 + rng.random(wts[k].shape)).astype(np.float32)
        m.set_weights(wts)
    PG = M.Params(); PD = M.Params()
    for k, v in gen.get_weights().items():
        PG[k] = v.astype(np.float64)
    for k, v in disc.get_weights().items():
        PD[k] = v.astype(np.float64)
    nf_merge = 2 * dcfg['n_filters']
    ocfg = dict(dcfg, lr_size=(h, w))
    g0, d0 = gen.get_weights(), disc.get_weights()

    def draw(seed):
        r = np.random.default_rng(seed)
        lr = r.random((B,) + lead + (h, w, n_ch)).astype(np.float32)
        st = r.random((B, H, W, 1)).astype(np.float32)
        hr = r.random((B,) + lead + (H, W, 1)).astype(np.float32)
        mask = (r.random((2 * B, nf_merge)) > 0.4).astype(np.float32)
        return lr, st, hr, mask, _cgan_banded(gkind, gcfg, g0, ocfg, d0, lr, hr, st, mask)
    lr, st, hr, mask, bref = first_quiet(draw, 90)           # (inputs on which the oracle's reference is well defined)
    PGt, PDt = M.convert(PG, T, requires_grad=True), M.convert(PD, T, requires_grad=True)
    t64 = lambda a: T.asarray(a.astype(np.float64))
    ref = TR.cgan_step(gkind, gcfg, PGt, ocfg, PDt, t64(lr), t64(hr), t64(st),
                       dropout_masks=(t64(mask[:B]), t64(mask[B:])))
    assert set(PD.keys()) == set(ref['gradsD'].keys())
    eng = CGANEngine(gen, disc, loss='mae', learning_rate=2e-4, beta_1=0.5)
    out = eng.step([lr, st], hr, dropout_keep=mask, apply_update=False)
    for i, k in enumerate(('gen_total', 'gen_gan', 'gen_px', 'disc')):
        assert out[i] == pytest.approx(ref[k], rel=1e-4), k
    gg, gd = gen.get_gradients(), disc.get_gradients()
    assert_matches_reference({'G:' + k: v for k, v in gg.items()} | {'D:' + k: v for k, v in gd.items()}, bref,
                             what=(gkind, dcfg))
    if dcfg.get('normalization') == 'bn':
        # moving averages after the step: updated by the real batch, then by the generated one (cgan.py:599-600)
        wd = disc.get_weights()
        assert ref['bn_updates']
        for name, (mm, mv) in ref['bn_updates'].items():
            np.testing.assert_allclose(wd[name + '/moving_mean'], mm.numpy(), rtol=2e-4, atol=2e-6, err_msg=name)
            np.testing.assert_allclose(wd[name + '/moving_variance'], mv.numpy(), rtol=2e-4, atol=2e-6, err_msg=name)
            assert np.abs(gd[name + '/moving_mean']).max() == 0.0


# ------------------------------------------------------------------------------------------------ block variants (f2)
def _oracle_pass(kind, ocfg, P, x, s, y, loss, ctx_kw, noises, training=True, dtype=np.float64):
    """Training-mode oracle pass (torch, fp64 unless told otherwise) with the dropout noise the device drew: loss, grads,
    prediction, ctx."""
    ctx = M.Ctx(training=training, noises=None if noises is None else [n.astype(dtype) for n in noises], **ctx_kw)
    PT = M.convert(P, T, dtype=dtype, requires_grad=True)
    pred = M.MODELS[kind](T, PT, T.asarray(x.astype(dtype)), None if s is None else T.asarray(s.astype(dtype)),
                          ctx=ctx, **ocfg)
    lv = TR.LOSSES[loss](T.asarray(y.astype(dtype)), pred)
    keys = [k for k in PT if not k.endswith(('moving_mean', 'moving_variance'))]
    gs = torch.autograd.grad(lv, [PT[k] for k in keys], allow_unused=True)
    grads = {k: (np.zeros(PT[k].shape) if g is None else g.numpy()) for k, g in zip(keys, gs)}
    return float(lv.detach()), grads, pred.detach().numpy(), ctx


def _banded_pass(kind, ocfg, P, x, s, y, loss, ctx_kw, noises):
    """tests/parity.py's reference (mid-point of the +/- band evaluations, band, fp32 noise floor) of _oracle_pass."""
    return banded_reference(lambda dt: _oracle_pass(kind, ocfg, P, x, s, y, loss, ctx_kw, noises, dtype=dt)[:3])


VARIANT_CASES = [
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, n_blocks=2, n_filters=8),
     dict(normalization='ln'), (2, 12, 16, 2), (2, 24, 32, 1)),
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, n_blocks=2, n_filters=8, attention=True),
     dict(normalization='bn'), (3, 12, 16, 1), None),
    ('net_postupsampling', dict(backbone_block='convnet', upsampling='rc', scale=2, n_blocks=1, n_filters=6,
                                activation='elu'), dict(normalization='bn', dropout_rate=0.2), (2, 10, 14, 3), None),
    ('net_postupsampling', dict(backbone_block='densenet', upsampling='spc', scale=2, n_blocks=2, n_filters=8),
     dict(normalization='ln', dropout_rate=0.3, dropout_variant='spatial'), (2, 10, 14, 1), (2, 20, 28, 2)),
    ('net_pin', dict(backbone_block='densenet', n_blocks=1, n_filters=8), dict(normalization='bn'), (2, 16, 16, 2), None),
    ('net_pin', dict(backbone_block='resnet', n_blocks=2, n_filters=8),
     dict(dropout_rate=0.25, dropout_variant='gaussian'), (2, 16, 16, 2), None),
    ('net_pin', dict(backbone_block='convnet', n_blocks=2, n_filters=8), dict(dropout_rate=0.5, dropout_variant='mcdrop'),
     (2, 16, 16, 1), (2, 16, 16, 1)),
    ('unet_pin', dict(n_filters=4, n_blocks=2, decoder_upsampling='spc'),
     dict(normalization='bn', dropout_rate=0.2, dropout_variant='spatial'), (2, 16, 24, 2), None),
    ('recnet_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=2, time_window=2, n_filters=4,
                                   n_blocks=1), dict(normalization='ln', dropout_rate=0.3, dropout_variant='spatial'),
     (2, 2, 10, 6, 2), None),
    ('recnet_pin', dict(backbone_block='convnet', time_window=3, n_filters=4, n_blocks=1),
     dict(normalization='bn', dropout_rate=0.2), (2, 3, 8, 8, 2), None),
    # ConvNext backbone (sp_postups.py:120-131,193-210): 7x7 stem, depthwise 7x7 + LN(1e-6) + Dense/act/Dense blocks,
    # ConvNextBlock aux branch, 7x7 closing ConvBlocks
    ('net_postupsampling', dict(backbone_block='convnext', upsampling='spc', scale=2, n_blocks=2, n_filters=8),
     dict(normalization='ln'), (2, 12, 16, 2), (2, 24, 32, 1)),
    ('net_postupsampling', dict(backbone_block='convnext', upsampling='rc', scale=2, n_blocks=1, n_filters=4,
                                activation='gelu'), dict(normalization='bn', dropout_rate=0.2), (2, 10, 14, 1), None),
    ('net_pin', dict(backbone_block='convnext', n_blocks=2, n_filters=8), dict(normalization='ln'), (2, 16, 16, 3),
     (2, 16, 16, 2)),
]


@pytest.mark.parametrize('kind,cfg,var,xs,ss', VARIANT_CASES)
def test_normalization_and_dropout_variants(kind, cfg, var, xs, ss):
    """Builders with normalization= / dropout_rate= / dropout_variant= (blocks.py:63-103,210-277,380-398,679-706):
    training-mode forward, loss and gradients against the oracle fed with the SAME dropout noise (read back from the
    device), BatchNormalization moving averages after the step, and the inference-mode forward afterwards."""
    from dl4ds_amd.training import SupervisedEngine
    model, P, ocfg = build_pair(kind, cfg, xs, ss, ctx_kw=var)
    B = xs[0]
    g = model.graph
    eng = SupervisedEngine(model, loss='mae', learning_rate=1e-3)
    w_before = model.get_weights()
    rate = var.get('dropout_rate', 0)

    def attempt(seed):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal(xs).astype(np.float32)
        s = None if ss is None else rng.standard_normal(ss).astype(np.float32)
        inputs = [x] if s is None else [x, s]
        probe = M.Ctx(training=True, **var)        # the oracle declares how many noise arrays it consumes and their shapes
        ref_shape = M.MODELS[kind](N, P, x.astype(np.float64), None if s is None else s.astype(np.float64), ctx=probe,
                                   **ocfg).shape
        y = rng.standard_normal(ref_shape).astype(np.float32)
        model.set_weights(w_before)                          # (an earlier attempt has moved the BatchNormalization averages)
        l_hip, g_hip = eng.loss_and_grads(inputs, y)
        assert g.dropout_count() == len(probe.noise_shapes)
        noises = [g.dropout_mask(i, B).reshape(shp) for i, shp in enumerate(probe.noise_shapes)]
        return x, s, inputs, y, l_hip, g_hip, noises, probe, _banded_pass(kind, ocfg, P, x, s, y, 'mae', var, noises)
    # inputs (and the device's own dropout draw) for which the oracle's reference is well defined: tests/parity.first_quiet
    x, s, inputs, y, l_hip, g_hip, noises, probe, bref = first_quiet(attempt, 21)
    for nz in noises:                                        # the noise has the statistics the layer promises
        if set(np.unique(nz)) <= {0.0, 1.0}:                 # keep mask (ConvBlock_att always uses plain Dropout)
            if nz.size > 2000:
                assert abs(nz.mean() - (1 - rate)) < 0.05
        else:
            assert var.get('dropout_variant') in ('gaussian', 'mcgaussiandrop')
            if nz.size > 2000:
                assert abs(nz.mean() - 1) < 0.05 and abs(nz.std() - np.sqrt(rate / (1 - rate))) < 0.05
    lv, grads, pred, ctx = _oracle_pass(kind, ocfg, P, x, s, y, 'mae', var, noises)
    assert l_hip == pytest.approx(lv, rel=1e-4)
    assert_matches_reference(g_hip, bref, what=(kind, var))
    w_after = model.get_weights()
    for k in w_after:
        if k.endswith(('moving_mean', 'moving_variance')):
            lname = k.rsplit('/', 1)[0]
            if lname in ctx.bn_updates:
                upd = ctx.bn_updates[lname][0 if k.endswith('moving_mean') else 1].detach().numpy()
                np.testing.assert_allclose(w_after[k], upd, rtol=2e-4, atol=2e-6, err_msg=k)
            else:                                            # DenseBlock.norm1: variables only
                np.testing.assert_array_equal(w_after[k], w_before[k])
            assert np.abs(g_hip[k]).max() == 0.0             # never trained
        else:
            np.testing.assert_array_equal(w_after[k], w_before[k])
    # inference: moving statistics, dropout off unless MC
    P2 = M.Params()
    for k, v in w_after.items():
        P2[k] = v.astype(np.float64)
    mc = (var.get('dropout_variant') or '').startswith('mc')
    out = model(inputs, training=False)
    noises2 = [g.dropout_mask(i, B).reshape(shp) for i, shp in enumerate(probe.noise_shapes)] if mc else None
    ctx2 = M.Ctx(training=False, noises=noises2, **var)
    ref = M.MODELS[kind](N, P2, x.astype(np.float64), None if s is None else s.astype(np.float64), ctx=ctx2, **ocfg)
    assert rel(out, ref) < 1e-3
    if mc:
        assert any(not np.array_equal(a, b) for a, b in zip(noises, noises2))      # a fresh draw per call
    if not mc:
        assert eng.evaluate(inputs, y) == pytest.approx(float(np.abs(ref - y).mean()), rel=1e-4)


def test_injected_dropout_mask_is_used_once():
    import dl4ds_amd.models as PM
    m = PM.net_pin('convnet', 1, 0, (8, 8), n_blocks=1, n_filters=4, dropout_rate=0.5, seed=1)
    g = m.graph
    x = np.random.default_rng(0).standard_normal((2, 8, 8, 1)).astype(np.float32)
    m(x, training=True)
    keep = [np.ones_like(g.dropout_mask(i, 2)) for i in range(g.dropout_count())]
    for i, k in enumerate(keep):
        g.set_dropout_mask(i, 2, k)
    a = m(x, training=True)                      # all-ones keep mask: the inference output scaled through 1/(1-p) chains
    for i, k in enumerate(keep):
        np.testing.assert_array_equal(g.dropout_mask(i, 2), k)
    b = m(x, training=True)                      # the next call draws again
    assert not np.array_equal(a, b)
    assert any(not np.array_equal(g.dropout_mask(i, 2), keep[i]) for i in range(len(keep)))


def _random_combo(i):
    """Deterministic sample of the post-/pre-upsampling builders' argument space (seed = case index)."""
    r = np.random.default_rng(1000 + i)
    pick = lambda xs: xs[int(r.integers(len(xs)))]
    kind = pick(['net_postupsampling', 'net_postupsampling', 'net_pin', 'unet_pin'])
    norm = pick([None, None, 'bn', 'ln'])
    drop = pick([0, 0, 0.2])
    var = dict(normalization=norm, dropout_rate=drop,
               dropout_variant=pick([None, 'gaussian', 'spatial', 'mcdrop']) if drop else None)
    aux = bool(r.integers(2))
    if kind == 'unet_pin':
        cfg = dict(n_filters=int(pick([4, 8])), n_blocks=int(pick([1, 2])), decoder_upsampling=pick(['spc', 'rc', 'dc']),
                   attention=bool(r.integers(2)), localcon_layer=bool(r.integers(2)))
        xs = (2, int(pick([16, 19])), int(pick([16, 22])), int(pick([1, 3])))
        ss = (2, xs[1], xs[2], 1) if aux else None
        return kind, cfg, var, xs, ss
    backbone = pick(['resnet', 'densenet', 'convnet', 'convnext'])
    if backbone == 'convnext' and norm is None:
        var['normalization'] = 'ln'
    cfg = dict(backbone_block=backbone, n_blocks=int(pick([1, 2])), n_filters=int(pick([4, 8])),
               attention=bool(r.integers(2)), localcon_layer=bool(r.integers(2)),
               activation=pick(['relu', 'relu', 'elu', 'gelu']), output_activation=pick([None, None, 'sigmoid']))
    h, w = int(pick([8, 10])), int(pick([8, 12]))
    if kind == 'net_postupsampling':
        ups = pick(['spc', 'rc', 'dc'])
        scale = int(pick([2, 4] if ups == 'spc' else [2]))
        cfg.update(upsampling=ups, scale=scale)
        xs = (2, h, w, int(pick([1, 2])))
        ss = (2, h * scale, w * scale, int(pick([1, 2]))) if aux else None
    else:
        xs = (2, 2 * h, 2 * w, int(pick([1, 2])))
        ss = (2, 2 * h, 2 * w, 1) if aux else None
    return kind, cfg, var, xs, ss


def _random_rec_combo(i):
    r = np.random.default_rng(5000 + i)
    pick = lambda xs: xs[int(r.integers(len(xs)))]
    kind = pick(['recnet_postupsampling', 'recnet_pin'])
    drop = pick([0, 0.2])
    var = dict(normalization=pick([None, 'bn', 'ln']), dropout_rate=drop,
               dropout_variant=pick([None, 'gaussian', 'spatial']) if drop else None)
    T = int(pick([2, 3]))
    cfg = dict(backbone_block=pick(['convnet', 'resnet', 'densenet']), time_window=T, n_filters=4, n_blocks=1,
               attention=bool(r.integers(2)), localcon_layer=bool(r.integers(2)), activation=pick(['relu', 'elu']))
    h, w = int(pick([6, 8])), int(pick([6, 10]))
    aux = bool(r.integers(2))
    if kind == 'recnet_postupsampling':
        cfg.update(upsampling=pick(['spc', 'rc', 'dc']), scale=2)
        return kind, cfg, var, (2, T, h, w, int(pick([1, 2]))), (2, 2 * h, 2 * w, 1) if aux else None
    return kind, cfg, var, (2, T, 2 * h, 2 * w, int(pick([1, 2]))), (2, 2 * h, 2 * w, 1) if aux else None


@pytest.mark.parametrize('i', range(28 + 10))
def test_random_builder_combinations(i):
    """28 spatial + 10 spatio-temporal seeded draws from backbone x upsampling x normalization x dropout variant x attention x localized convolution x
    auxiliary input x activations: training-mode loss and every gradient against the oracle fed the device's noise."""
    from dl4ds_amd.training import SupervisedEngine
    kind, cfg, var, xs, ss = _random_combo(i) if i < 28 else _random_rec_combo(i - 28)
    var = {k: v for k, v in var.items() if v not in (None, 0)}
    model, P, ocfg = build_pair(kind, cfg, xs, ss, ctx_kw=var or None)
    eng = SupervisedEngine(model, loss='mse', learning_rate=1e-3)
    g = model.graph

    def attempt(seed):
        rng = np.random.default_rng(seed)
        x = rng.standard_normal(xs).astype(np.float32)
        s = None if ss is None else rng.standard_normal(ss).astype(np.float32)
        inputs = [x] if s is None else [x, s]
        probe = M.Ctx(training=True, **var)
        ref_shape = M.MODELS[kind](N, P, x.astype(np.float64), None if s is None else s.astype(np.float64), ctx=probe,
                                   **ocfg).shape
        y = rng.standard_normal(ref_shape).astype(np.float32)
        l_hip, g_hip = eng.loss_and_grads(inputs, y)
        assert g.dropout_count() == len(probe.noise_shapes)
        noises = [g.dropout_mask(k, xs[0]).reshape(shp) for k, shp in enumerate(probe.noise_shapes)]
        return x, s, y, l_hip, g_hip, noises, _banded_pass(kind, ocfg, P, x, s, y, 'mse', var, noises)
    x, s, y, l_hip, g_hip, noises, bref = first_quiet(attempt, 77 + 100 * i)
    lv, grads, pred, ctx = _oracle_pass(kind, ocfg, P, x, s, y, 'mse', var, noises)
    assert l_hip == pytest.approx(lv, rel=2e-4), (kind, cfg, var)
    assert_matches_reference(g_hip, bref, what=(kind, cfg, var))


def _fusion_report(model, B):
    import ctypes, json
    from dl4ds_amd import _lib
    buf = ctypes.create_string_buffer(1 << 14)
    _lib.check(_lib.lib().dl4ds_graph_fusion_report(model.graph.h, B, buf, len(buf)))
    return json.loads(buf.value.decode())


@pytest.mark.parametrize('kind,kw', [
    ('net_postupsampling', dict(backbone_block='resnet', upsampling='spc', scale=4, lr_size=(12, 10), n_blocks=2)),
    ('net_postupsampling', dict(backbone_block='convnet', upsampling='rc', scale=2, lr_size=(24, 20), n_blocks=1, attention=True)),
    ('net_pin', dict(backbone_block='resnet', hr_size=(40, 36), n_blocks=2, attention=True)),
])
def test_attention_handed_to_neighbouring_convolutions_equals_separate_passes(monkeypatch, kind, kw):
    """ChannelAttention2D between two convolutions (ConvBlock_att -> ConvBlock_out, sp_postups.py:204-211): pooling taken
    from the producing convolution's epilogue, scale applied in the consumer's loads, dX = dY * scale + dmean applied in the
    producer's backward loads -- against the same model with DL4DS_NO_TAIL_FUSION=1 (three separate passes): same forward
    (the products are the same fp32 operations; only the pooling's summation order differs), same loss, gradients and
    three Adam steps to fp32 rounding.  The report must show that the hand-over really happened."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    build = getattr(PM, kind)
    cfg = dict(n_channels=2, n_aux_channels=0, n_filters=8, seed=5, **kw)
    monkeypatch.delenv('DL4DS_NO_TAIL_FUSION', raising=False)
    fused = build(**cfg)
    B = 3
    rep = _fusion_report(fused, B)
    tail = rep[-1]                                   # ConvBlock_att's attention is the last one of the graph
    assert tail['pool_from_producer'] and tail['scale_in_consumer_load'] and tail['dx_in_producer_backward'], rep
    assert tail['dscale_from_consumer_wgrad'], rep
    monkeypatch.setenv('DL4DS_NO_TAIL_FUSION', '1')
    plain = build(**cfg)
    assert not any(r['pool_from_producer'] or r['scale_in_consumer_load'] or r['dx_in_producer_backward']
                   for r in _fusion_report(plain, B))
    rng = np.random.default_rng(3)
    w = plain.get_weights()
    for k in w:
        if k.endswith('bias'):
            w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    plain.set_weights(w); fused.set_weights(w)
    xs = (B,) + fused.input_shapes[0]
    x = rng.standard_normal(xs).astype(np.float32)
    y = rng.standard_normal((B,) + fused.output_shape).astype(np.float32)
    assert rel(fused(x), plain(x)) < 2e-6
    ef = SupervisedEngine(fused, loss='mse', learning_rate=1e-3)
    ep = SupervisedEngine(plain, loss='mse', learning_rate=1e-3)
    lf, gf = ef.loss_and_grads([x], y)
    lp, gp = ep.loss_and_grads([x], y)
    assert lf == pytest.approx(lp, rel=1e-6)
    assert_grads_close(gf, gp, tol=1e-4, what='handed vs separate')
    # a smaller batch after a larger one re-uses the buffers laid out for the larger (fusion addresses must stay valid)
    assert rel(fused(x[:1]), plain(x[:1])) < 2e-6
    for _ in range(3):
        lf, lp = ef.step([x], y), ep.step([x], y)
        assert lf == pytest.approx(lp, rel=1e-5)
    wf, wp = fused.get_weights(), plain.get_weights()
    for k in wf:
        assert np.abs(wf[k] - wp[k]).max() < 2e-4, k          # three Adam steps of 1e-3 each; sign flips of ~0 gradients aside


@pytest.mark.parametrize('kind,kw,ins', [
    ('unet_pin', dict(backbone_block='unet', n_channels=3, n_aux_channels=1, n_filters=8, n_blocks=3, hr_size=(32, 48), decoder_upsampling='dc'), 2),
    ('unet_pin', dict(backbone_block='unet', n_channels=2, n_aux_channels=0, n_filters=8, n_blocks=2, hr_size=(32, 32), decoder_upsampling='spc'), 1),
    ('net_postupsampling', dict(backbone_block='densenet', upsampling='spc', scale=2, n_channels=2, n_aux_channels=0, lr_size=(16, 12),
                                n_filters=8, n_blocks=2), 1),
])
def test_concatenate_without_copies_equals_concatenate_with_copies(monkeypatch, kind, kw, ins):
    """Concatenate inputs that live inside the concatenation's buffer -- activations (forward) and gradients (backward: the
    decoder levels of the U-Net, whose Concatenate then copies nothing) -- against the same model with DL4DS_NO_CONCAT_ALIAS=1
    (dense tensors, forward and backward copies): same forward, loss, gradients and three Adam steps to fp32 rounding."""
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import SupervisedEngine
    build = getattr(PM, kind)
    monkeypatch.delenv('DL4DS_NO_CONCAT_ALIAS', raising=False)
    monkeypatch.delenv('DL4DS_NO_GRAD_ALIAS', raising=False)
    aliased = build(seed=5, **kw)
    monkeypatch.setenv('DL4DS_NO_CONCAT_ALIAS', '1')
    dense = build(seed=5, **kw)
    monkeypatch.delenv('DL4DS_NO_CONCAT_ALIAS')
    rng = np.random.default_rng(3)
    w = dense.get_weights()
    for k in w:
        if k.endswith('bias'):
            w[k] = (rng.standard_normal(w[k].shape) * 0.1).astype(np.float32)
    dense.set_weights(w); aliased.set_weights(w)
    B = 3
    xs = [rng.standard_normal((B,) + s).astype(np.float32) for s in aliased.input_shapes[:ins]]
    y = rng.standard_normal((B,) + aliased.output_shape).astype(np.float32)
    assert rel(aliased(xs), dense(xs)) < 2e-6
    ea = SupervisedEngine(aliased, loss='mse', learning_rate=1e-3)
    ed = SupervisedEngine(dense, loss='mse', learning_rate=1e-3)
    la, ga = ea.loss_and_grads(xs, y)
    ld, gd = ed.loss_and_grads(xs, y)
    assert la == pytest.approx(ld, rel=1e-6)
    assert_grads_close(ga, gd, tol=1e-4, what='aliased vs copied')
    for _ in range(3):
        la, ld = ea.step(xs, y), ed.step(xs, y)
        assert la == pytest.approx(ld, rel=2e-5)
