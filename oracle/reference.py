"""The oracle's REFERENCE for a train step and the per-tensor gradient criterion (test infrastructure: used by tests/,
__graft_entry__.smoke() and nothing else; the product never imports oracle/).

THE GRADIENT CRITERION.  Every parameter's gradient is compared at ITS OWN scale (a single global scale lets a tensor
whose gradients are 100x smaller than the largest one be 10 % wrong and still pass):

    |g_hip - g_ref|_inf  <=  tol * |g_ref|_inf  +  ULPS * eps32 * gscale  +  band_k  +  k_noise * noise_k

* ``tol`` = 1e-3, north_star's tolerance, applied to tensor k's own largest entry.
* ``ULPS * eps32 * gscale`` (64 x 6e-8 x the largest gradient entry of the whole model = 3.8e-6 gscale): single precision
  cannot resolve a result below a few units in the last place of the quantities it was computed from.  It matters for
  tensors whose exact gradient is ZERO or nearly so -- e.g. everything upstream of a LayerNormalization over ONE channel
  (ConvBlock_out with normalization='ln': dx = rstd * (g dy - mean(g dy)) is exactly 0, in fp32 a rounding residual times
  rstd = 1 / sqrt(1e-3) = 32, observed 2 ... 15 ulps of gscale) -- and is 0.4 % of a tensor 1000x smaller than the largest.
* ``band_k``: the gradient of a ReLU network is a discontinuous function -- a pre-activation within rounding distance of
  zero takes either branch depending on summation order, and every upstream gradient moves by that unit's whole
  contribution (two fp32 evaluations of cfg1 at 128 x 128 whose forward values agree to 4e-7 differ by 1e-3 ... 5e-3 of
  a gradient tensor's size for that reason alone; tools/diag_parity.py).  The oracle therefore evaluates the gradient
  TWICE in fp64, with every derivative discontinuity (ReLU thresholds, hard-sigmoid clip points, the sign of the MAE
  residual, max-pooling ties) displaced by +BAND and by -BAND relative to the magnitude of its argument
  (oracle/torch_ops.py: KINK); the reference is the mid-point and ``band_k`` the spread |g+ - g-|_inf of tensor k -- zero
  whenever nothing lies within BAND of a discontinuity.  BAND = 2e-6 is >= 4x the forward error observed between the HIP
  path and the oracle (2e-7 ... 5e-7 of the output scale at the BASELINE sizes).
* ``noise_k`` = |g_fp32 - g_ref|_inf, the deviation of the oracle's OWN single-precision evaluation (torch CPU, other
  summation orders): the cancellation noise of that tensor.  It only matters for gradients that are sums of
  random-signed terms cancelling to ~1e-7 of their parts (e.g. LayerNormalization over a single channel).
Each term is computed by the oracle from the same inputs; nothing is fitted to the results under test.

``breakdown`` says, per tensor, how much of the bound each term supplied and whether the tensor passes on ``tol`` + ulp
floor ALONE -- the tests write it to profiles/parity_r04.json and assert on it, so the slack is visible, not implicit.

``oracle_reference``: evaluates the above for a supervised step or a CGAN step, sample by sample (the losses are batch
means, the models carry no batch statistics) -- in-process for small cases, over worker PROCESSES (oracle/worker.py)
at the BASELINE sizes, where one fp64 sample takes 1-3 s.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

BAND = 2e-6
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(v):
    return v.detach().numpy() if hasattr(v, 'detach') else np.asarray(v)


EPS32 = float(np.finfo(np.float32).eps) / 2          # unit round-off of single precision, 6e-8
ULPS = 64.0


def grad_failures(got, ref, tol=1e-3, ulps=ULPS, band=None, noise=None, k_noise=8.0):
    """-> [(name, err, bound)] of the tensors that violate the per-tensor criterion (empty = pass)."""
    refs = {k: _np(v).astype(np.float64) for k, v in ref.items() if v is not None}
    gscale = max((np.abs(v).max() for v in refs.values() if v.size), default=0.0)
    bad = []
    for k, r in refs.items():
        g = np.asarray(got[k], np.float64)
        assert g.shape == r.shape, (k, g.shape, r.shape)
        err = float(np.abs(g - r).max()) if r.size else 0.0
        bound = tol * (float(np.abs(r).max()) if r.size else 0.0) + ulps * EPS32 * gscale
        if band is not None:
            bound += band[k]
        if noise is not None:
            bound += k_noise * noise[k]
        if not err <= bound:
            bad.append((k, err, bound))
    return bad


def assert_grads_close(got, ref, tol=1e-3, ulps=ULPS, what='', band=None, noise=None):
    bad = grad_failures(got, ref, tol, ulps, band, noise)
    assert not bad, (what, [(k, f'{e:.3e} > {b:.3e}') for k, e, b in bad[:8]], len(bad))


def assert_matches_reference(got, ref, key='grads', tol=1e-3, what=''):
    """``ref``: what oracle_reference returned; ``key``: 'grads' | 'gradsG' | 'gradsD'."""
    sfx = key[5:]
    assert_grads_close(got, ref[key], tol=tol, what=what, band=ref['band' + sfx], noise=ref['noise' + sfx])
    return slack_report(ref, key)


def slack_report(ref, key='grads'):
    """How much the oracle's own slack (band + noise floor) grants each tensor, relative to that tensor's size (tensors
    below 1e-3 of the model's largest gradient are measured against that level: they live on the ulp floor anyway).
    -> [(fraction, name)] sorted, largest first.  The slack must stay a correction to the 1e-3 criterion, not become it."""
    sfx = key[5:]
    gscale = max(float(np.abs(v).max()) for v in ref[key].values() if v.size)
    rows = [((ref['band' + sfx][k] + 8.0 * ref['noise' + sfx][k]) / max(float(np.abs(v).max()), 1e-3 * gscale), k)
            for k, v in ref[key].items() if v.size]
    return sorted(rows, reverse=True)


def breakdown(got, ref, key='grads', tol=1e-3, ulps=ULPS, k_noise=8.0):
    """Per tensor: the error and every term of its bound, all relative to the tensor's OWN largest entry (tensors below 1e-3 of
    the model's largest gradient are measured against that level, as in slack_report).
    -> [dict(name, size, err, tol, ulp, band, noise, plain_ok, ok)]: ``plain_ok`` = passes on tol + ulp floor alone."""
    sfx = key[5:]
    refs = {k: _np(v).astype(np.float64) for k, v in ref[key].items() if v is not None}
    gscale = max((np.abs(v).max() for v in refs.values() if v.size), default=0.0)
    rows = []
    for k, r in refs.items():
        if not r.size:
            continue
        g = np.asarray(got[k], np.float64)
        own = float(np.abs(r).max())
        scale = max(own, 1e-3 * gscale) or 1.0
        err = float(np.abs(g - r).max())
        t_tol, t_ulp = tol * own, ulps * EPS32 * gscale
        t_band, t_noise = ref['band' + sfx][k], k_noise * ref['noise' + sfx][k]
        rows.append(dict(name=k, size=int(r.size), err=err / scale, tol=t_tol / scale, ulp=t_ulp / scale, band=t_band / scale,
                         noise=t_noise / scale, plain_ok=bool(err <= t_tol + t_ulp), ok=bool(err <= t_tol + t_ulp + t_band + t_noise)))
    return rows


def targets_clear_of_the_kink(pred, rng):
    """Targets for the gradient comparisons: prediction +/- (0.5 ... 1.5), the sign + for 80 % of the pixels.  |pred - y| >= 0.5
    keeps every MAE residual far from its sign change, and the mostly-coherent sign makes the parameter gradients sums that
    do NOT cancel (with pure-noise targets every gradient entry is a random walk over the pixels: one ReLU unit on the other
    side of its threshold then moves it by ~2 / sqrt(#pixels), percent-level, in any single-precision evaluation -- the
    oracle's own band shows it).  The field of signs is still random pixel by pixel, so the upstream gradient entering the
    last layers varies from pixel to pixel."""
    s = np.where(rng.random(pred.shape) < 0.8, 1.0, -1.0)
    return (pred.astype(np.float64) + s * (0.5 + rng.random(pred.shape))).astype(np.float32)


def oracle_params(weights):
    from oracle import models as M
    P = M.Params()
    for k, v in weights.items():
        P[k] = np.asarray(v, np.float64)
    return P


def _job(what, kind, cfg, weights, x, s, y, loss, dcfg, dweights, mask, band):
    B = int(x.shape[0])
    meta = dict(what=what, kind=kind, cfg=cfg, loss=loss, B=B, band=band)
    job = dict(x=np.asarray(x, np.float32), y=np.asarray(y, np.float32))
    if s is not None:
        job['s'] = np.asarray(s, np.float32)
    if what == 'supervised':
        meta['w_names'] = list(weights)
        for k, v in weights.items():
            job['w/' + k] = np.asarray(v, np.float32)
    else:
        meta['dcfg'] = dcfg
        meta['g_names'], meta['d_names'] = list(weights), list(dweights)
        for k, v in weights.items():
            job['g/' + k] = np.asarray(v, np.float32)
        for k, v in dweights.items():
            job['d/' + k] = np.asarray(v, np.float32)
        job['mask'] = np.asarray(mask, np.float32)
    return job, meta


def _merge(parts, meta, names_by_prefix):
    B = meta['B']
    tot = {}
    for part in parts:
        for k in (part.files if hasattr(part, 'files') else part):
            if k in ('pred', 'pred_idx'):
                continue
            tot[k] = np.asarray(part[k], np.float64) if k not in tot else tot[k] + part[k]
    out = {}
    for sfx, names in names_by_prefix.items():
        gp = {k: tot['p' + sfx + '/' + k] for k in names}
        gm = {k: tot['m' + sfx + '/' + k] for k in names}
        g32 = {k: tot['f32' + sfx + '/' + k] for k in names}
        mid = {k: 0.5 * (gp[k] + gm[k]) for k in names}
        out['grads' + sfx] = mid
        out['band' + sfx] = {k: float(np.abs(gp[k] - gm[k]).max()) if gp[k].size else 0.0 for k in names}
        out['noise' + sfx] = {k: float(np.abs(g32[k] - mid[k]).max()) if mid[k].size else 0.0 for k in names}
    lp, lm = tot['loss/p'], tot['loss/m']
    out['losses'] = 0.5 * (lp + lm)
    out['loss_spread'] = float(np.abs(lp - lm).max())
    out['loss'] = float(out['losses'][0])
    if any('pred' in (p.files if hasattr(p, 'files') else p) for p in parts):
        shp = None
        for p in parts:
            if 'pred' in (p.files if hasattr(p, 'files') else p):
                shp = p['pred'].shape[1:]
        pred = np.zeros((B,) + tuple(shp), np.float64)
        for p in parts:
            if 'pred' in (p.files if hasattr(p, 'files') else p):
                pred[np.asarray(p['pred_idx'])] = p['pred']
        out['pred'] = pred
    return out


def banded_reference(call, band=BAND):
    """The same reference for ANY oracle evaluation: ``call(dtype)`` -> (losses, {name: gradient}, prediction) is run
    under +band and -band in fp64 and once in fp32 (whole batch at once: batch statistics and batch-wide losses allowed).
    -> dict(loss, losses, loss_spread, pred, grads, band, noise)."""
    from oracle import torch_ops as T
    res = {}
    for tag, shift, dt in (('p', +band, np.float64), ('m', -band, np.float64), ('f32', 0.0, np.float32)):
        with T.kink_shift(shift):
            lv, g, pred = call(dt)
        res[tag] = (np.atleast_1d(np.asarray(lv, np.float64)), {k: _np(v).astype(np.float64) for k, v in g.items() if v is not None},
                    None if pred is None else _np(pred).astype(np.float64))
    names = list(res['p'][1])
    mid = {k: 0.5 * (res['p'][1][k] + res['m'][1][k]) for k in names}
    return dict(grads=mid,
                band={k: float(np.abs(res['p'][1][k] - res['m'][1][k]).max()) if mid[k].size else 0.0 for k in names},
                noise={k: float(np.abs(res['f32'][1][k] - mid[k]).max()) if mid[k].size else 0.0 for k in names},
                losses=0.5 * (res['p'][0] + res['m'][0]), loss=float(0.5 * (res['p'][0][0] + res['m'][0][0])),
                loss_spread=float(np.abs(res['p'][0] - res['m'][0]).max()),
                pred=None if res['p'][2] is None else plain_forward(call))


def plain_forward(call):
    """Prediction of ``call`` in fp64 with NO displacement (the displaced passes' forward values sit ~1e-5 off)."""
    from oracle import torch_ops as T
    with T.kink_shift(0.0):
        return _np(call(np.float64)[2]).astype(np.float64)


def oracle_reference(what, kind, cfg, weights, x, s, y, loss='mae', dcfg=None, dweights=None, mask=None, band=BAND,
                     workers=0):
    """The oracle's reference for one supervised step (``what='supervised'``: -> loss, pred, grads, band, noise) or one
    CGAN step (``what='cgan'``: x = conditioning array, y = HR array; -> losses[4], gradsG/bandG/noiseG,
    gradsD/bandD/noiseD).  ``workers`` > 0: that many worker processes share the samples."""
    assert loss in ('mae', 'mse'), 'sample-by-sample evaluation needs a loss that is a batch mean'
    job, meta = _job(what, kind, cfg, weights, x, s, y, loss, dcfg, dweights, mask, band)
    names = {'': list(weights)} if what == 'supervised' else {'G': list(weights), 'D': list(dweights)}
    B = meta['B']
    if workers <= 0:
        from oracle.worker import run_job
        return _merge([run_job(job, meta, list(range(B)))], meta, names)
    workers = min(workers, B)
    ncpu = os.cpu_count() or 8
    env = dict(os.environ, ORACLE_WORKER_THREADS=str(max(1, min(16, ncpu // workers))), PYTHONPATH=ROOT)
    with tempfile.TemporaryDirectory() as tmp:
        jp = os.path.join(tmp, 'job.npz')
        np.savez(jp, meta=np.asarray(json.dumps(meta)), **job)
        procs = []
        for i in range(workers):
            op = os.path.join(tmp, f'out{i}.npz')
            procs.append((subprocess.Popen([sys.executable, os.path.join(ROOT, 'oracle', 'worker.py'), jp, str(i),
                                            str(workers), op], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT), op))
        parts = []
        for p, op in procs:
            log, _ = p.communicate()
            assert p.returncode == 0, f'oracle worker failed:\n{log.decode()[-2000:]}'
            parts.append({k: v for k, v in np.load(op).items()})
        return _merge(parts, meta, names)
