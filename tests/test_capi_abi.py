"""The C-ABI library builds, loads, and exports every symbol include/peclr_hip.h declares.
No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

from tests.conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "peclr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(peclr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    syms = declared_symbols()
    for s in ("peclr_version", "peclr_gemm_f32", "peclr_bn_relu_fwd_f32", "peclr_bn_relu_bwd_f32",
              "peclr_align_fwd_f32", "peclr_align_bwd_f32", "peclr_ntxent_fwd_f32", "peclr_ntxent_bwd_f32",
              "peclr_ntxent_finalize_f32", "peclr_lars_sumsq_f32", "peclr_lars_adam_update_f32"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from peclr_amd import _capi

    path = _capi.library_path()
    if not os.path.exists(path):
        import __graft_entry__ as ge

        ge.build()
    handle = ctypes.CDLL(path)
    for s in declared_symbols():
        assert hasattr(handle, s), f"{s} is declared in peclr_hip.h but not exported"
    assert set(_capi.SIGNATURES) == set(declared_symbols())
    assert _capi.lib().peclr_version() >= 100
    assert _capi.lib().peclr_error_string(-4) == b"workspace too small"


def test_argument_errors_without_gpu():
    """Argument validation happens before any launch, so it is checkable on CPU."""
    from peclr_amd import _capi

    L = _capi.lib()
    assert L.peclr_gemm_f32(0, 4, 4, 4, None, 4, None, 4, None, 4, None, 1, None, None) == -1
    assert L.peclr_gemm_pick_split_k(256, 512, 2048) == 8
    assert L.peclr_gemm_pick_split_k(256, 128, 512) == 8
    assert L.peclr_gemm_pick_split_k(4096, 4096, 64) == 1
    assert L.peclr_ntxent_jsplit(256, 256, 0) == 8 and L.peclr_ntxent_jsplit(256, 256, 1) == 8
    assert L.peclr_ntxent_jsplit(256, 2048, 0) == 64     # 8-GPU config: local rows x global columns
    assert L.peclr_ntxent_jsplit(0, 256, 0) == 0
    assert L.peclr_ntxent_fwd_f32(None, 8, 0, None, 8, 64, 4, 2.0, None, None, None, 1, None) == -1
    assert L.peclr_align_fwd_f32(None, 1, 4, 128, 2, 0, None, None, None, None, 1.0, 1.0, None, None, None, None,
                                 None, None, None) == -1


def test_product_fails_loudly_on_cpu_tensors():
    import torch

    from peclr_amd import _capi, ops

    z = torch.nn.functional.normalize(torch.randn(8, 128))
    with pytest.raises(_capi.PeclrHipError, match="no CPU path"):
        ops.ntxent(z, 4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "peclr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_integration_stub_matches_the_abi():
    """The ctypes stub shown to a reference maintainer in INTEGRATION.md declares the same argument lists
    as the library's own binding."""
    import re
    from ctypes import c_float, c_int, c_void_p

    from peclr_amd import _capi
    from tests.conftest import ROOT

    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    kinds = {"_P": c_void_p, "_I": c_int, "_F": c_float}
    stubs = re.findall(r"_L\.(\w+)\.argtypes = \[([^\]]*)\]", text)
    assert len(stubs) >= 5
    for name, args in stubs:
        assert [kinds[a.strip()] for a in args.split(",")] == _capi.SIGNATURES[name][1], name
    for name in _capi.SIGNATURES:                       # every entry point is documented there
        assert name in text or name in ("peclr_version", "peclr_error_string"), name
