"""Shared helpers of the GPU parity tests (not a test module).

The oracle's reference for a train step and the per-tensor gradient criterion live in oracle/reference.py (so that
__graft_entry__.smoke() can use them without importing tests/); this module re-exports them and adds what needs the
library: ``kernel_tags`` = the set of kernel tags the library launched while a callable ran (dl4ds_profile_*), so a test
can assert that the kernels it means to check are the ones that were dispatched, and ``record`` = the collector behind
profiles/parity_r06.json.
"""
import ctypes
import json
import os

import numpy as np

from oracle.reference import (BAND, EPS32, ULPS, ROOT, assert_grads_close, banded_reference, breakdown, grad_failures,  # noqa: F401
                              oracle_params, oracle_reference, plain_forward, slack_report, targets_clear_of_the_kink)

PARITY_LOG = {}          # what -> summary of oracle.reference.breakdown; tests/conftest.py writes it out at session end

# ---- the caps every oracle comparison is held to (VERDICT r4 "give the parity tests teeth everywhere") ---------------------------
# The criterion itself is per ELEMENT (oracle/reference.py).  On top of it the EVIDENCE must stay evidence:
ELEMENT_PLAIN_FRACTION = 0.99    # share of all gradient entries of a comparison that pass on north_star's 1e-3 (+ ulp floor) ALONE
TENSOR_SLACK_FRACTION = 0.01     # no tensor may have more than this share of its entries pass only through their band / noise term
# Numeric per-case overrides, all in this one table: str(what) -> dict(elements=.., tensor=.., entries=..) replacing the two caps above /
# granting tensors with fewer than 1 / TENSOR_SLACK_FRACTION entries that many slack-reliant entries (a 4-entry bias cannot have "1 %").
CAP_OVERRIDES = {
}
REPORT_ONLY = os.environ.get('DL4DS_PARITY_REPORT_ONLY', '') == '1'       # collect the artefact without asserting the caps (tuning runs)


# ---- inputs on which the reference is well defined --------------------------------------------------------------------------
# The gradient of these networks is discontinuous wherever a ReLU / hard-sigmoid / max-pool / |.| argument sits on its kink, and on
# the small grids of the model tests (a few hundred pixels per gradient sum) one such unit moves whole filters by percents.  The
# oracle itself says whether a set of inputs has such a unit: its two displaced evaluations then disagree (band_i > 0).  Tests
# that can draw their inputs freely take the FIRST seed of a fixed list for which the oracle grants no entry more than
# QUIET_LIMIT of its tensor's size -- a selection made from the oracle alone, before any result of the HIP path is looked at --
# and are then held to the caps like every other comparison (with nothing granted, every entry has to pass on 1e-3 itself).
QUIET_LIMIT = 5e-4
QUIET_SEEDS = 8


def is_quiet(ref, limit=QUIET_LIMIT):
    return all(slack_report(ref, key)[0][0] < limit for key in ('grads', 'gradsG', 'gradsD') if key in ref)


def first_quiet(attempt, first_seed, n=QUIET_SEEDS, ref_of=lambda out: out[-1]):
    """``attempt(seed)`` -> tuple whose last item is the oracle's reference (see ``ref_of``).  Returns attempt's result for the
    first of the seeds first_seed, first_seed + 1, ... whose reference is quiet; the last attempt's if none is."""
    out = None
    for seed in range(first_seed, first_seed + n):
        out = attempt(seed)
        if is_quiet(ref_of(out)):
            break
    return out


def record(what, rows, full=False):
    """Summary of a comparison (and, for the BASELINE-size ones, every tensor's row) for the session's parity artefact."""
    entry = summarize(rows)
    if full:
        entry['tensors'] = [{k: (round(v, 9) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
    else:                                   # the tensors that drew on their allowance are always listed
        entry['tensors_on_slack'] = [{k: (round(v, 9) if isinstance(v, float) else v) for k, v in r.items()} for r in rows if r['n_slack'] or r.get('n_ulp', 0)]
    PARITY_LOG[str(what)] = entry


def assert_caps(rows, what=''):
    """The two statements about the evidence (see the constants above), with the numeric overrides of CAP_OVERRIDES."""
    ov = CAP_OVERRIDES.get(str(what), {})
    el_cap, t_cap, small = ov.get('elements', ELEMENT_PLAIN_FRACTION), ov.get('tensor', TENSOR_SLACK_FRACTION), ov.get('entries', 0)
    sm = summarize(rows)
    assert sm['element_plain_frac'] >= el_cap, f'{what}: only {sm["element_plain_frac"]:.4f} of the entries pass on 1e-3 alone: {sm}'
    for r in rows:
        allowed = max(int(t_cap * r['size']), small)
        assert r['n_slack'] <= allowed, f'{what}: {r["n_slack"]} of {r["size"]} entries of {r["name"]} pass only through band / noise (cap {allowed}): {r}'
    return sm


def assert_matches_reference(got, ref, key='grads', tol=1e-3, what='', full=False, caps=True):
    """``ref``: what oracle_reference returned; ``key``: 'grads' | 'gradsG' | 'gradsD'.  Asserts the per-element criterion AND the
    caps on how much of the comparison may rest on the oracle's slack, records the per-tensor breakdown under ``what``
    (``full``: with every tensor's row) and returns it."""
    sfx = key[5:]
    rows = breakdown(got, ref, key, tol)
    if what:
        record(what, rows, full)
    assert_grads_close(got, ref[key], tol=tol, what=what, band=ref['band' + sfx], noise=ref['noise' + sfx])
    if caps and not REPORT_ONLY:
        assert_caps(rows, what)
    return rows


def summarize(rows):
    """-> dict of a breakdown: tensors (n, plain_ok, plain_frac), entries (elements, elements_on_slack, element_plain_frac), the
    largest allowance granted (worst_slack) / drawn on (worst_slack_used), the largest error, the tensor with the largest share
    of slack-reliant entries."""
    n = len(rows)
    ok = sum(r['plain_ok'] for r in rows)
    ws = max(rows, key=lambda r: r['band'] + r['noise'])
    we = max(rows, key=lambda r: r['err'])
    wf = max(rows, key=lambda r: r['n_slack'] / r['size'])
    ne, ns = sum(r['size'] for r in rows), sum(r['n_slack'] + r['n_bad'] for r in rows)
    return dict(n=n, plain_ok=ok, plain_frac=ok / max(n, 1), elements=ne, elements_on_slack=ns, element_plain_frac=1.0 - ns / max(ne, 1),
                elements_on_ulp_floor=sum(r.get('n_ulp', 0) for r in rows), tensors_on_ulp_floor=sum(1 for r in rows if r.get('n_ulp', 0)),
                worst_slack=ws['band'] + ws['noise'], worst_slack_name=ws['name'], worst_slack_used=max(r['slack_used'] for r in rows),
                worst_err=we['err'], worst_err_name=we['name'],
                worst_tensor_slack_frac=wf['n_slack'] / wf['size'], worst_tensor_slack_name=wf['name'], worst_tensor_slack_entries=wf['n_slack'])


def kernel_tags(fn):
    """Run ``fn()`` with the library's per-launch profiler on; -> (fn's result, {tag: launches})."""
    import dl4ds_amd._lib as L
    lib = L.lib()
    L.check(lib.dl4ds_profile_filter(b''))
    L.check(lib.dl4ds_profile_enable(1))
    try:
        out = fn()
        buf = ctypes.create_string_buffer(1 << 17)
        L.check(lib.dl4ds_profile_report(buf, len(buf)))
        rep = json.loads(buf.value.decode())
    finally:
        L.check(lib.dl4ds_profile_enable(0))
    return out, {k: v['n'] for k, v in rep.items()}
