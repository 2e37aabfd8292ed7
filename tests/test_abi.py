"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/mmdistill.h declares."""
import ctypes
import os
import re

import pytest


def test_library_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from mm_distillnet_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    dll = ctypes.CDLL(_lib.LIB_PATH)
    names = re.findall(r"\bint\s+(mmd_\w+)\s*\(", open(_lib.HEADER).read())
    assert len(names) >= 40
    for n in names:
        assert hasattr(dll, n), n
    sigs = _lib.LIB.symbols()
    assert set(sigs) == set(names)
    assert dll.mmd_pp_cap() == 1024
    # (round 6: the second build with the bf16-storage branches and its seven *_w16 entry points were deleted with the bf16_hbm mode)
    assert not [n for n in names if n.endswith("_w16")] and not hasattr(_lib, "LIB16")


def test_bad_arguments_are_rejected_without_gpu():
    from mm_distillnet_amd import _lib
    dll = _lib.LIB.load()
    # argument validation happens before any launch: null pointers / bad sizes -> -22
    assert dll.mmd_pwconv_fwd(None, None, None, 0, 0, 0, None, None, 0, None, None, None, 0, None, 0, None, None, None, 0, None, None, 0, 0, None, 0, None) == -22
    assert dll.mmd_dwconv_fwd(None, None, None, 1, 8, 8, 16, 4, 1, None, None, 0, None, None, None, 0, None, None, 0, None, None, None, 0, None) == -22
    with pytest.raises(ValueError):
        import torch
        _lib.call("mmd_colsum", torch.zeros(4, 4), torch.zeros(4), 4, 4)      # host tensor: refused loudly


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mm_distillnet_amd import _lib
    fresh = _lib._Lib(str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        fresh.load()
