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
