"""CPU-side checks of the boundary: the built library loads and exports every symbol that
include/dl4ds_hip.h declares (no compute calls -- there is no GPU here)."""
import ctypes
import os
import pytest

import dl4ds_amd._lib as L


def test_header_parses_and_library_exports_every_symbol():
    protos = L.parse_header()
    assert len(protos) > 60
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = L.load()
    for name in protos:
        assert hasattr(lib, name), f'{name} declared in include/dl4ds_hip.h but not exported'


def test_last_error_is_a_string():
    lib = L.load()
    msg = lib.dl4ds_last_error()
    assert isinstance(msg, bytes)


def test_product_does_not_import_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'dl4ds_amd')
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dp, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f
