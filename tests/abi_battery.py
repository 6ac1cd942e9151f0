"""Host-side argument battery of the C ABI (no GPU needed: every call here must be answered by the entry point's own argument
checks or size arithmetic, before any launch).  Run as a script by tests/test_abi.py, once against the shipped library and once
against the AddressSanitizer + UndefinedBehaviorSanitizer twin (lib/asan, with the ASan runtime preloaded):

    python tests/abi_battery.py <path to libdgdm_hip.so>

Prints "battery ok: <n> calls" and exits 0; a sanitizer report or a crash ends the process with another status."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dgdm_histopath_lab_amd import _lib  # noqa: E402

OK_CODES = (0, -1, -2, -3)        # DGDM_OK / INVALID_ARG / UNSUPPORTED / WORKSPACE: never -4 (a launch was attempted)


def value(t, mode):
    if t is C.c_void_p:
        return None
    if t is C.c_float:
        return (0.0, -1.0, 1e30, 0.5)[mode]
    if t in (C.c_int32, C.c_int):
        return (0, -1, 2 ** 31 - 1, 7)[mode]
    if t is C.c_uint32:
        return (0, 1, 2 ** 32 - 1, 7)[mode]
    if t is C.c_int64:
        return (0, -1, 2 ** 62, 7)[mode]
    if t is C.c_size_t:
        return (0, 1, 2 ** 62, 7)[mode]
    raise TypeError(t)


def main():
    lib = _lib.open_library(sys.argv[1])
    n = 0
    # every entry point: all pointers null, scalars all zero / all negative-or-one / all huge / all small-odd
    for name, (res, args) in _lib.SIGNATURES.items():
        if name == "dgdm_error_string":
            continue
        for mode in range(4):
            r = getattr(lib, name)(*[value(t, mode) for t in args])
            n += 1
            if res is C.c_int and args and args[-1] is C.c_void_p:        # a status-returning entry point (takes a stream last)
                if name in ("dgdm_seed_epoch_advance", "dgdm_seed_epoch_set"):   # no arguments to refuse: they do launch (-4 without a GPU)
                    assert r in OK_CODES + (-4,), (name, mode, r)
                else:
                    assert r in OK_CODES, (name, mode, r)
    for code in range(-6, 2):
        assert lib.dgdm_error_string(code)
        n += 1
    # workspace / planning functions over a sweep of sizes, including the largest representable ones
    big = [1, 2, 3, 63, 64, 65, 255, 256, 4095, 4096, 40000, 10 ** 6, 2 ** 24, 2 ** 30, 2 ** 31 - 1]
    for m in big:
        for nn in (1, 64, 128, 2 ** 20, 2 ** 31 - 1):
            for k in (1, 16, 768, 2 ** 31 - 1):
                for fn in ("dgdm_gemm_tn_chunks", "dgdm_gemm_tn_chunks_grouped"):
                    c = getattr(lib, fn)(m, nn, k)
                    assert 1 <= c <= max(1, m), (fn, m, nn, k, c)
                for fn in ("dgdm_gemm_tn_workspace_bytes", "dgdm_gemm_tn_bf16x3_workspace_bytes", "dgdm_gemm_tn_f16x2_workspace_bytes"):
                    getattr(lib, fn)(m, nn, k, 1)
                n += 5
        lib.dgdm_csr_build_workspace_bytes(m * 5, m, 1); lib.dgdm_csr_build_pair_workspace_bytes(m * 5, m, 1)
        lib.dgdm_csr_build_pair_status_offset(m * 5, m, 1)
        lib.dgdm_spmm_long_item_cap(m * 6); lib.dgdm_spmm_long_slot_cap(m * 6); lib.dgdm_spmm_long_table_words(m * 6)
        lib.dgdm_topk_perm_workspace_bytes(m); lib.dgdm_pool_score_bwd_workspace_bytes(m, 64)
        lib.dgdm_rownorm_bwd_workspace_bytes(m, 512, 1); lib.dgdm_rownorm_bwd_slots(m, 512, 8)
        lib.dgdm_segment_sum_workspace_bytes(min(m, 4096), 128)
        lib.dgdm_gemm_image_bytes(m, 768); lib.dgdm_gemm_image_blocks(m, 768)
        lib.dgdm_attn_pack_bytes(m, 16, 0); lib.dgdm_attn_pack_bytes(m, 16, 3)
        n += 15
    # descriptor arrays read on the host: AdamW
    T = _lib.AdamTensor
    step, ticket = C.c_float(0.0), C.c_uint32(0)      # never dereferenced: every array below is rejected (or empty) before a launch
    sp, tp = C.cast(C.pointer(step), C.c_void_p), C.cast(C.pointer(ticket), C.c_void_p)

    def adam(entries, count=None):
        arr = (T * max(len(entries), 1))(*[T(*e) for e in entries])
        return lib.dgdm_adamw_step(C.cast(arr, C.c_void_p), len(entries) if count is None else count, None, 1e-3, 0.9, 0.999, 1e-8, 0.01, sp, tp, None)
    assert adam([]) == 0                                                       # nothing to do
    assert adam([(0, 0, 0, 0, 0), (0, 0, 0, 0, 0)]) == 0                       # empty tensors only: no launch
    assert adam([(4096, 4096, 4096, 4096, -1)]) == -1                          # negative size
    assert adam([(0, 4096, 4096, 4096, 8)]) == -1                              # null parameter
    assert adam([(4097, 4096, 4096, 4096, 8)]) == -2                           # misaligned pointer
    assert adam([(4096, 4096, 4096, 4096, 8)], count=-3) == -1
    assert lib.dgdm_adamw_step(None, 2, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, sp, tp, None) == -1
    assert lib.dgdm_adamw_step(None, 0, None, 1e-3, 1.5, 0.999, 1e-8, 0.0, sp, tp, None) == -1     # beta1 out of range
    n += 8
    # ... the many-problem dW launch and its reduction
    P, R = _lib.TnPartial, _lib.TnReduce
    parr = (P * 2)(P(None, None, None, None, None, 0, 0, 0, 0, 0, 0, 0), P(4096, 4096, 4096, 4096, 4096, 8, 8, 16, -5, 3, 3, 0))
    assert lib.dgdm_gemm_tn_partial_many_f16x2(C.cast(parr, C.c_void_p), 2, None) == -1
    assert lib.dgdm_gemm_tn_partial_many_f16x2(C.cast(parr, C.c_void_p), _lib.TN_PARTIAL_MAX + 1, None) == -1
    assert lib.dgdm_gemm_tn_partial_many_f16x2(None, 0, None) == 0
    rarr = (R * 2)(R(None, None, None, None, 0, 0, 0, 0, 0, 0), R(4096, 4096, None, None, 3, 0, -1, 4, 4, 4))
    assert lib.dgdm_gemm_tn_reduce_many(C.cast(rarr, C.c_void_p), 2, None) in (-1, -2)
    assert lib.dgdm_gemm_tn_reduce_many(None, 0, None) == 0
    n += 5
    # ... the long-row table descriptor of the gather
    L = _lib.LongRows
    lr = L(None, None, 0, 0, 0)
    assert lib.dgdm_spmm(4096, 4096, 4096, 4096, 8, 4, 4096, 8, 4, 8, None, 0, C.byref(lr), None) == -1
    lr = L(4096, 4096, 4, 16, 16)                                                # scratch narrower than C
    assert lib.dgdm_spmm(4096, 4096, 4096, 4096, 8, 4, 4096, 8, 4, 8, None, 0, C.byref(lr), None) == -3
    n += 2
    print(f"battery ok: {n} calls")


if __name__ == "__main__":
    main()
