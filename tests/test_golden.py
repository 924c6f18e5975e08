"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py with the fp64 torch
oracle).  CPU: the independent numpy backend of the oracle reproduces them.  GPU: the HIP path reproduces
forward, loss and every parameter-gradient norm within the 1e-3 relative tolerance of north_star."""
import os

import numpy as np
import pytest

from oracle import np_ops as N
from oracle import models as M
from tests.golden.make_golden import CASES, golden_weights

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _load(name):
    return np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False)


def _oracle_params(case):
    c = CASES[case]
    P0 = M.init_params(c['model'], (1,) + c['x'][1:], None if c['s'] is None else (1,) + c['s'][1:], **c['cfg'])
    P = M.Params()
    for k, v in P0.items():
        P[k] = golden_weights(k, v.shape).astype(np.float64)
    return P


@pytest.mark.parametrize('case', sorted(CASES))
def test_numpy_oracle_reproduces_golden(case):
    c, g = CASES[case], _load(case)
    P = _oracle_params(case)
    s = g['s'].astype(np.float64) if 's' in g.files else None
    pred = M.MODELS[c['model']](N, P, g['x'].astype(np.float64), s, **c['cfg'])
    np.testing.assert_allclose(pred, g['pred'], rtol=2e-4, atol=2e-5)
    assert getattr(N, c['loss'])(g['y_true'].astype(np.float64), pred) == pytest.approx(float(g['loss']), rel=1e-6)


def _product_model(case):
    import dl4ds_amd.models as PM
    c = CASES[case]
    x, s = c['x'], c['s']
    kw = dict(n_channels=x[-1], n_aux_channels=0 if s is None else s[-1])
    if c['model'] == 'net_pin':
        m = PM.net_pin(hr_size=x[1:3], **kw, **c['cfg'])
    elif c['model'] == 'net_postupsampling':
        m = PM.net_postupsampling(lr_size=x[1:3], **kw, **c['cfg'])
    elif c['model'] == 'recnet_postupsampling':
        m = PM.recnet_postupsampling(lr_size=x[2:4], **kw, **c['cfg'])
    elif c['model'] == 'unet_pin':
        m = PM.unet_pin('unet', hr_size=x[1:3], **kw, **c['cfg'])
    m.set_weights({k: golden_weights(k, v.shape) for k, v in m.get_weights().items()})
    return m


@pytest.mark.gpu
@pytest.mark.parametrize('case', sorted(CASES))
def test_hip_reproduces_golden(case):
    from dl4ds_amd.training import SupervisedEngine
    c, g = CASES[case], _load(case)
    m = _product_model(case)
    inputs = [g['x']] + ([g['s']] if 's' in g.files else [])
    pred = m(inputs)
    scale = np.abs(g['pred']).max()
    assert np.abs(pred - g['pred']).max() / scale < 1e-3
    eng = SupervisedEngine(m, loss=c['loss'])
    lv, grads = eng.loss_and_grads(inputs, g['y_true'])
    assert lv == pytest.approx(float(g['loss']), rel=1e-3)
    names = [str(n) for n in g['grad_names']]
    assert sorted(grads.keys()) == names
    gmax = g['grad_norms'].max()
    for i, k in enumerate(names):
        # every tensor at its own scale (the fixture holds each gradient's l2 norm and first four entries)
        ref_n = g['grad_norms'][i]
        assert abs(np.linalg.norm(grads[k].astype(np.float64)) - ref_n) <= 1e-3 * max(ref_n, 1e-6 * gmax), k
        head = grads[k].ravel()[:4]
        rms = ref_n / np.sqrt(grads[k].size)
        assert np.abs(head - g['grad_heads'][i][:head.size]).max() <= 1e-3 * max(np.abs(g['grad_heads'][i]).max(), rms, 1e-6 * gmax), k


@pytest.mark.gpu
def test_hip_cgan_step_reproduces_golden():
    import dl4ds_amd.models as PM
    from dl4ds_amd.training import CGANEngine
    g = _load('cfg5_cgan_step')
    H = 32
    gen = PM.unet_pin('unet', 5, 1, hr_size=(H, H), n_filters=8, n_blocks=3, decoder_upsampling='dc')
    disc = PM.residual_discriminator(5, 'pin', False, 8, (H // 8, H // 8), n_filters=8, n_res_blocks=2, hr_size=(H, H))
    gen.set_weights({k: golden_weights('G/' + k, v.shape) for k, v in gen.get_weights().items()})
    disc.set_weights({k: golden_weights('D/' + k, v.shape) for k, v in disc.get_weights().items()})
    eng = CGANEngine(gen, disc, loss='mae')
    out = eng.step([g['lr'], g['st']], g['hr'], dropout_keep=g['mask'], apply_update=False)
    np.testing.assert_allclose(out, g['losses'], rtol=1e-3)
    gg, gd = gen.get_gradients(), disc.get_gradients()
    for names, norms, got in ((g['g_names'], g['g_norms'], gg), (g['d_names'], g['d_norms'], gd)):
        for k, ref in zip(names, norms):
            assert abs(np.linalg.norm(got[str(k)].astype(np.float64)) - ref) <= 1e-3 * max(ref, 1e-6 * norms.max()), k
