"""CPU: the C-ABI shared library loads and exports every symbol include/sgpo_hip.h declares
(no compute calls: there is no GPU here); error paths that need no device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'sgpo_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(sgpo_[a-z0-9_]+)\s*\(', src)))


def test_header_symbols_exported_and_bound():
    from sgpo_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f'{n} declared in include/sgpo_hip.h but not exported'
        assert n in _lib.SIGNATURES, f'{n} has no ctypes signature in sgpo_amd/_lib.py'
    assert lib.sgpo_version() >= 1
    assert lib.sgpo_prof_kind_name(0).decode().startswith('gemm')


def test_argument_errors_without_device():
    from sgpo_amd import _lib
    lib = _lib.load()
    cfg = _lib.DitConfig(31, 512, 128, 256, 8, 8, 2048, 1024)
    n = lib.sgpo_dit_param_count(ctypes.byref(cfg))
    assert n == 28579871 + 0 or n > 28_000_000          # 28.58 M parameters
    rc = lib.sgpo_dit_create(ctypes.byref(cfg), None, n, None)
    assert rc != 0 and b'null' in lib.sgpo_last_error()
    rc = lib.sgpo_reverse_step_mdlm(None, None, None, None, 0, 1, 1, 31, 28, 0.5, 0.1, 0.0, 0, 0, None, 0, 0, 0, None,
                                    None, None, None)
    assert rc != 0
    bn = _lib.BytenetConfig(31, 8, 1024, 512, 16, 5, 128)
    assert lib.sgpo_bytenet_param_count(ctypes.byref(bn)) > 37_000_000   # "38M" ByteNet


def test_missing_library_fails_loudly(monkeypatch):
    from sgpo_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libsgpo_hip.so')
    with pytest.raises(_lib.SgpoError):
        _lib.load()


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from sgpo_amd import _lib, runtime
    with pytest.raises(_lib.SgpoError):
        runtime.init()
