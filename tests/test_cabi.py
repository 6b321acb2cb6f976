"""The C-ABI library loads and exports every symbol include/adaisp.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ada(?:isp|yolo)_\w+)\s*\(", text)))


def test_libadaisp_exports_header_symbols():
    from adaptiveisp_amd import _lib
    L = _lib.load()
    names = _declared("adaisp.h")
    assert set(_lib.EXPORTS) <= set(names)
    for n in names:
        assert hasattr(L, n), f"libadaisp.so does not export {n}"
    assert L.adaisp_abi_version() == _lib.ABI_VERSION
    assert L.adaisp_strerror(0) == b"ok" and b"alias" in L.adaisp_strerror(-3)
    expect = {-1: 0, 0: 1, 1: 1, 2: 9, 3: 1, 4: 1, 5: 8, 6: 1, 7: 1, 8: 1, 9: 3, 10: 2, 11: 1, 12: 24, 13: -1}
    for op, n in expect.items():
        assert L.adaisp_num_params(op) == n


def test_argument_checks_without_gpu():
    from adaptiveisp_amd import _lib
    L = _lib.load()
    # null pointers / bad sizes are rejected before anything touches a device
    assert L.adaisp_process(0, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_forward(None, None, None, None, None, 1, 1, 8, 8, 0, None) == -1
    assert L.adaisp_pool64(None, None, 1, 8, 8, None) == -1
    buf = (ctypes.c_float * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.adaisp_process(99, p, p, p, 1, 1, 2, 2, 0, None) == -2          # unknown op
    assert L.adaisp_process(2, p, p, p, 1, 1, 2, 2, 0, None) == -1           # CCM needs 9 params, stride 1
    assert L.adaisp_process(3, p, p, p, 1, 1, 2, 2, 0, None) == -3           # stencil in place
