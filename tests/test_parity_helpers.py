"""The gradient criterion's own machinery (tests/parity.py), checked on CPU: the discontinuity band is what it claims to
be, worker processes and the in-process path agree, and the criterion still rejects a wrong gradient."""
import numpy as np
import pytest
import torch

from oracle import models as M
from oracle import torch_ops as T
from tests.parity import BAND, banded_reference, grad_failures, oracle_reference

KIND = 'net_postupsampling'
CFG = dict(backbone_block='resnet', upsampling='spc', scale=2, n_blocks=1, n_filters=4)


def _case(B=3):
    P0 = M.init_params(KIND, (1, 8, 8, 1), None, dtype=np.float32, **CFG)
    w = {k: np.asarray(v) for k, v in P0.items()}
    rng = np.random.default_rng(0)
    return w, rng.standard_normal((B, 8, 8, 1)).astype(np.float32), rng.standard_normal((B, 16, 16, 1)).astype(np.float32)


def test_worker_processes_and_in_process_reference_agree():
    w, x, y = _case()
    a = oracle_reference('supervised', KIND, CFG, w, x, None, y)
    b = oracle_reference('supervised', KIND, CFG, w, x, None, y, workers=2)
    assert a['loss'] == pytest.approx(b['loss'], rel=1e-14)
    np.testing.assert_allclose(a['pred'], b['pred'], rtol=0, atol=1e-15)
    for k in a['grads']:
        np.testing.assert_allclose(a['grads'][k], b['grads'][k], rtol=0, atol=1e-16)
        np.testing.assert_allclose(a['band'][k], b['band'][k], rtol=1e-9, atol=1e-15 * np.abs(a['grads'][k]).max() + 1e-30)
        assert a['band'][k].shape == a['grads'][k].shape == a['noise'][k].shape          # per ELEMENT since round 5
    # no unit of this small case lies within BAND of a kink: the band is the smooth O(BAND) change only, the fp32 noise
    # floor is single-precision rounding -- both far inside the 1e-3 criterion
    for k, g in a['grads'].items():
        if np.abs(g).max() > 0:
            assert a["band"][k].max() < 5e-4 * np.abs(g).max(), k
            assert a['noise'][k].max() < 1e-4 * np.abs(g).max(), k


def test_band_brackets_a_unit_on_the_discontinuity():
    """y = relu(w * x) with one input exactly at the kink: the +band evaluation switches the unit off, the -band evaluation on;
    the band equals that unit's whole gradient contribution and is zero without it."""
    def ref_for(xv):
        x = np.asarray(xv, np.float64)

        def call(dt):
            wt = torch.tensor(np.asarray([1.0], dt), requires_grad=True)
            out = T.relu(wt * torch.tensor(x.astype(dt)))
            loss = out.sum()
            return float(loss), {'w': torch.autograd.grad(loss, wt)[0]}, out.detach()
        return banded_reference(call)
    tied = ref_for([2.0, 1e-9, -1.0])            # |x| max = 2: the kink moves by +/- 4e-6
    assert tied['band']['w'][0] == pytest.approx(1e-9, rel=1e-6)     # d/dw of the tied unit = its x
    clear = ref_for([2.0, 0.5, -1.0])
    assert clear['band']['w'][0] == 0.0
    assert clear['grads']['w'][0] == pytest.approx(2.5)


def test_criterion_still_rejects_wrong_gradients():
    w, x, y = _case()
    ref = oracle_reference('supervised', KIND, CFG, w, x, None, y)
    good = {k: v.astype(np.float32) for k, v in ref['grads'].items()}
    assert not grad_failures(good, ref['grads'], band=ref['band'], noise=ref['noise'])
    k = 'ResidualBlock1/conv1/kernel'
    bad = dict(good)
    bad[k] = good[k].copy()
    bad[k].flat[3] += 5e-3 * np.abs(good[k]).max()            # one entry off by 0.5 % of THIS tensor's size
    fails = grad_failures(bad, ref['grads'], band=ref['band'], noise=ref['noise'])
    assert [f[0] for f in fails] == [k]
    assert BAND == 2e-6


def test_band_is_granted_to_the_entries_next_to_the_kink_only():
    """one entry near a kink must not loosen the bound of its whole tensor.  y_j = relu(w_j * x_j) with entry 1 tied:
    an evaluation that takes the other branch of the tied unit passes on that entry's own band (its two displaced evaluations
    differ by the unit's whole contribution), the same deviation on the entry next to it fails, and the breakdown counts exactly
    one entry on slack.  The reference itself is the UNDISPLACED gradient: an evaluation that does not cross the kink needs no slack."""
    from tests.parity import assert_caps, breakdown
    x = np.asarray([2.0, 1e-9, -1.0, 0.75])             # (0.75: exact in fp32, so the noise term of that entry is zero)

    def call(dt):
        wt = torch.tensor(np.ones(4, dt), requires_grad=True)
        out = T.relu(wt * torch.tensor(x.astype(dt)))
        loss = out.sum()
        return float(loss), {'w': torch.autograd.grad(loss, wt)[0]}, out.detach()
    ref = banded_reference(call)
    assert ref['band']['w'].tolist() == pytest.approx([0.0, 1e-9, 0.0, 0.0])
    g = ref['grads']['w'].copy()                                  # the undisplaced gradient: [2, 1e-9, 0, 0.75] (the tied unit is on)
    assert g.tolist() == pytest.approx([2.0, 1e-9, 0.0, 0.75])
    on = g.copy(); on[1] = 0.0                                    # the tied unit switched OFF by a rounding error: inside its own band
    assert not grad_failures({'w': on}, ref['grads'], band=ref['band'], noise=ref['noise'], ulps=0.0, tol=1e-12)
    off = g.copy(); off[3] += 1e-9                                # the same deviation on an entry with no kink nearby
    fails = grad_failures({'w': off}, ref['grads'], band=ref['band'], noise=ref['noise'], ulps=0.0, tol=1e-12)
    assert [(f[0], f[3]) for f in fails] == [('w', 1)]
    rows = breakdown({'w': on}, dict(ref), tol=1e-12, ulps=0.0)
    assert rows[0]['n_slack'] == 1 and rows[0]['n_bad'] == 0 and not rows[0]['plain_ok'] and rows[0]['ok']
    with pytest.raises(AssertionError, match='of the entries pass on 1e-3 alone'):
        assert_caps(rows, 'unit case')                            # 1 of 4 entries on slack: over the 1 % cap


def test_noise_floor_is_the_tensors_median_not_one_sample():
    from oracle.reference import elementwise_slack
    rng = np.random.default_rng(0)
    g = rng.standard_normal(1000)
    g32 = g + 1e-6 * rng.standard_normal(1000)
    g32[5] = g[5]                                                # the oracle's fp32 sample happens to be exact here
    mid, band, noise = elementwise_slack(g, g, g32)
    assert band.max() == 0.0 and np.all(mid == g)
    sigma = 1.4826 * np.median(np.abs(g32 - g))
    assert noise[5] == pytest.approx(sigma) and 0.8e-6 < sigma < 1.2e-6
    assert noise.max() == np.abs(g32 - g).max()
