"""The C-ABI library loads without a GPU and exports every symbol include/vfr.h declares."""
import ctypes
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "vfr.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vfr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    from vfr_amd import _vfr
    _vfr.build()
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(str(_vfr.LIB_PATH))
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vfr.h but not exported by libvfr.so"
        assert n in _vfr.SIGNATURES, f"{n} has no ctypes signature in _vfr.SIGNATURES"
    assert set(_vfr.SIGNATURES) <= set(names)


def test_version_and_error_plumbing_without_gpu():
    from vfr_amd import _vfr
    l = _vfr.lib()
    assert l.vfr_version() == 100
    assert l.vfr_set_option(b"no-such-option", 1) == -1            # VFR_EINVAL, no GPU needed
    assert b"unknown option" in l.vfr_last_error()
    assert l.vfr_linear_f32(None, 1, 1, None, None, 1, 0, None, None) == -1
    assert l.vfr_score_topk_workspace_bytes(5000, 10000, 100) > 0
    assert l.vfr_bilstm_workspace_bytes(64, 20, 100, 1000, 400) > 0


def test_hip_wrappers_refuse_cpu_tensors():
    import pytest
    import torch
    from vfr_amd import _vfr
    with pytest.raises(RuntimeError, match="ROCm device"):
        _vfr.linear(torch.zeros(2, 2), torch.zeros(2, 2))
