"""Shared helpers of the GPU parity tests (not a test module).

The oracle's reference for a train step and the per-tensor gradient criterion live in oracle/reference.py (so that
__graft_entry__.smoke() can use them without importing tests/); this module re-exports them and adds what needs the
library: ``kernel_tags`` = the set of kernel tags the library launched while a callable ran (dl4ds_profile_*), so a test
can assert that the kernels it means to check are the ones that were dispatched, and ``record`` = the collector behind
profiles/parity_r04.json.
"""
import ctypes
import json
import os

import numpy as np

from oracle.reference import (BAND, EPS32, ULPS, ROOT, assert_grads_close, banded_reference, breakdown, grad_failures,  # noqa: F401
                              oracle_params, oracle_reference, plain_forward, slack_report, targets_clear_of_the_kink)

PARITY_LOG = {}          # what -> rows of oracle.reference.breakdown; tests/conftest.py writes it out at session end


def record(what, rows, full=False):
    """Summary of a comparison (and, for the BASELINE-size ones, every tensor's row) for the session's parity artefact."""
    entry = summarize(rows)
    if full:
        entry['tensors'] = [{k: (round(v, 9) if isinstance(v, float) else v) for k, v in r.items()} for r in rows]
    PARITY_LOG[str(what)] = entry


def assert_matches_reference(got, ref, key='grads', tol=1e-3, what='', full=False):
    """``ref``: what oracle_reference returned; ``key``: 'grads' | 'gradsG' | 'gradsD'.  Asserts the per-tensor criterion,
    records the per-tensor breakdown under ``what`` (``full``: with every tensor's row) and returns it."""
    sfx = key[5:]
    rows = breakdown(got, ref, key, tol)
    if what:
        record(what, rows, full)
    assert_grads_close(got, ref[key], tol=tol, what=what, band=ref['band' + sfx], noise=ref['noise' + sfx])
    return rows


def summarize(rows):
    """-> dict(n, plain_ok, plain_frac, worst_slack, worst_slack_name, worst_err, worst_err_name) of a breakdown."""
    n = len(rows)
    ok = sum(r['plain_ok'] for r in rows)
    ws = max(rows, key=lambda r: r['band'] + r['noise'])
    we = max(rows, key=lambda r: r['err'])
    return dict(n=n, plain_ok=ok, plain_frac=ok / max(n, 1), worst_slack=ws['band'] + ws['noise'], worst_slack_name=ws['name'],
                worst_err=we['err'], worst_err_name=we['name'])


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
