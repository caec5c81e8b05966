"""The C-ABI library loads without a GPU and exports every symbol include/ibhip.h declares."""
import ctypes
import os
import re

import pytest

import ibamd
from ibamd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "ibhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ibh_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_header_symbols():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ibhip.h but not exported"
    assert lib.ibh_version() >= 100


def test_binding_covers_header():
    assert set(_declared()) == set(_lib.EXPORTS)


def test_null_arguments_are_reported_not_crashed():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.ibh_partition_create(ctypes.byref(h), 2, 0, None, None, None, None, None, None, None, None, None, 0, None,
                                  None, 0, 0)
    assert rc != 0 and b"null" in lib.ibh_last_error()
    assert lib.ibh_partition_destroy(None) == 0


def test_compute_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    with pytest.raises(_lib.IbhError):
        ibamd.hip(np.zeros(4, dtype=np.float32))


def test_ew_eval_validates_its_program_on_the_host():
    """ibh_ew_eval checks the postfix program before anything is launched (no GPU needed to be told it is malformed)."""
    lib = _lib.load()
    PUSH_A, PUSH_S, ADD, ABS = 32, 33, 0, 16
    buf = (ctypes.c_float * 4)()
    arrs = (ctypes.c_void_p * 1)(ctypes.addressof(buf))
    nvs = (ctypes.c_int32 * 1)(1)
    sc = (ctypes.c_float * 1)(2.0)

    def run(prog, narr=1, nscal=1):
        P = (ctypes.c_int32 * len(prog))(*prog)
        return lib.ibh_ew_eval(0, 1, len(prog), P, narr, arrs, nvs, nscal, sc, ctypes.addressof(buf))

    assert run([PUSH_A, PUSH_S, ADD]) == 0                     # n = 0 rows: valid program, nothing to launch
    assert run([PUSH_A, ABS, PUSH_S | (0 << 8), ADD]) == 0
    for bad, what in (([ADD], b"two operands"), ([PUSH_A, PUSH_A], b"exactly one value"), ([PUSH_A | (3 << 8)], b"array"),
                      ([PUSH_S | (1 << 8)], b"scalar"), ([PUSH_A, 99], b"unknown"), ([PUSH_A] * 9 + [ADD] * 8, b"deeper"),
                      ([PUSH_A, ABS] * 25, b"48 instructions")):
        assert run(bad) != 0 and what in lib.ibh_last_error(), (bad, lib.ibh_last_error())
