"""CPU-side checks of the C-ABI library: it builds for gfx950, loads, and exports every symbol
include/dgdm_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    from dgdm_histopath_lab_amd import _build
    return _build.build(verbose=False)


def _declared():
    hdr = open(os.path.join(ROOT, "include", "dgdm_hip.h")).read()
    return sorted(set(re.findall(r"DGDM_API\s+[\w\s\*]+?\b(dgdm_\w+)\s*\(", hdr)))


def test_header_symbols_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in dgdm_hip.h but not exported"


def test_python_binding_matches_header(lib_path):
    from dgdm_histopath_lab_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    hdr = open(os.path.join(ROOT, "include", "dgdm_hip.h")).read()
    for name, (_, args) in _lib.SIGNATURES.items():
        m = re.search(r"DGDM_API[\w\s\*]+?\b%s\s*\(([^;]*?)\)\s*;" % name, hdr, re.S)
        assert m, name
        params = [p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"]
        assert len(params) == len(args), f"{name}: header has {len(params)} params, binding {len(args)}"


def test_host_side_argument_checks(lib_path):
    """Entry points validate arguments on the host before any launch (safe without a GPU)."""
    from dgdm_histopath_lab_amd import _lib
    lib = _lib.load()
    assert lib.dgdm_abi_version() >= 1
    assert lib.dgdm_error_string(0) == b"ok" and b"workspace" in lib.dgdm_error_string(-3)
    assert lib.dgdm_csr_build_workspace_bytes(50000, 10000, 1) > 4 * (2 * 60000 + 2 * 10000)
    assert lib.dgdm_csr_build(None, 10, 4, 1, 0, None, None, None, None, 0, None) == -1   # null pointers
    assert lib.dgdm_csr_build_pair_workspace_bytes(50000, 10000, 1) > 4 * (4 * 60000 + 4 * 10000)
    assert lib.dgdm_csr_build_pair(None, 10, 4, 1, None, None, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.dgdm_spmm(None, None, None, None, 8, 4, None, 8, 4, 8, None, 0, None) == -1
    assert lib.dgdm_spmm(1, 1, 1, 16, 6, 4, 16, 8, 4, 6, None, 0, None) == -2                 # C % 4 != 0
