"""The C-ABI library loads without a GPU and exports exactly what include/sober_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import torch

from sober_amd import _native as nat

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "sober_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t)\s+(sober_\w+)\s*\(", src)))


def test_header_and_binding_agree():
    decl = declared_functions()
    assert len(decl) >= 20
    assert sorted(nat.SIGNATURES) == decl


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name
    assert nat.load().sober_abi_version() == nat.ABI_VERSION


def test_host_only_entry_points():
    assert [nat.padded_dim(d) for d in (1, 2, 4, 5, 10, 13, 20, 21, 32)] == [4, 4, 4, 8, 12, 16, 20, 24, 32]
    assert [nat.bit_words(d) for d in (1, 64, 65, 128, 2048)] == [1, 1, 2, 2, 32]
    try:
        nat.padded_dim(33)
        assert False
    except nat.SoberHipError:
        pass
    lib = nat.load()
    assert lib.sober_level_chunks(700, 0, 100000, 200) >= 1
    assert lib.sober_level_chunks(700, 0, 100000, 200) <= 500
    assert lib.sober_level_chunks(700, 0, 150, 200) == 1
    assert lib.sober_level_chunks(0, 0, 10, 2) == -1
    # argument errors never reach the GPU
    assert lib.sober_dgemm(0, 0, 0, 1, 1, 1.0, None, 1, None, 1, 0.0, None, 1, None) == -1
    assert lib.sober_level_reduce(0, None, None, 1, None, None, 4, None, 0, 1, 1, None, None, 1.0, 1,
                                  None, 1, 0, None, 0, None) == -1


def test_car_pivot_host_handles_early_exit():
    """Q6: no positive entry in the pivot column -> stop (SOBER/_rchq.py:241-242)."""
    Phi = -torch.ones(6, 3, dtype=torch.double)
    mu = torch.full((6,), 1 / 6, dtype=torch.double)
    assert nat.car_pivot_host(Phi, mu) == 0
    assert np.allclose(mu.numpy(), 1 / 6)
