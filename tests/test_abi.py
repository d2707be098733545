"""CPU: the C-ABI library loads and exports every symbol include/ssdk.h declares (no compute without a GPU)."""
import ctypes
import os
import re

from conftest import REPO
from single_shot_detection_amd import _lib


def header_symbols():
    text = open(os.path.join(REPO, 'include', 'ssdk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(ssdk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'libssdk.so not built: run __graft_entry__.build()'
    handle = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 10
    missing = [s for s in syms if not hasattr(handle, s)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    assert set(header_symbols()) == set(_lib.exported_symbols())


def test_version_and_error_string():
    lib = _lib.lib()
    import re
    declared = int(re.search(r'#define\s+SSDK_VERSION\s+(\d+)', open(os.path.join(REPO, 'include', 'ssdk.h')).read()).group(1))
    assert lib.ssdk_version() == declared >= 111   # (the built library is the header's revision: a stale .so fails here)
    # invalid-argument path is host-only: no GPU needed
    assert lib.ssdk_linspace_f32(0.0, 1.0, 0, None) < 0
    assert b'ssdk_linspace_f32' in lib.ssdk_last_error_string()


def test_host_anchor_helpers_match_oracle():
    import numpy as np
    import oracle
    out = np.empty(7, np.float32)
    assert _lib.lib().ssdk_linspace_f32(0.15, 1.05, 7, out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(out.view(np.uint32), oracle.linspace_f32(0.15, 1.05, 7).view(np.uint32))
