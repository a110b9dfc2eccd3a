"""ctypes binding of libgpupoly.so (include/gpupoly.h).

This is the same binding surface the reference's Rust side declares in
`src/poly/dcrt/gpu.rs:69-240`.  The library is mandatory: importing any
compute-facing part of `mxx_amd` without the built HIP library raises — there is
no CPU fallback in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MXX_GPUPOLY_LIB: another build of the same library (same-box A/B runs of kernel variants: tools/ab_*.sh)
LIB_PATH = os.environ.get("MXX_GPUPOLY_LIB") or os.path.join(_HERE, "libgpupoly.so")

GPU_POLY_FORMAT_COEFF = 0
GPU_POLY_FORMAT_EVAL = 1
GPU_MATRIX_DIST_UNIFORM = 0
GPU_MATRIX_DIST_GAUSS = 1
GPU_MATRIX_DIST_BIT = 2
GPU_MATRIX_DIST_TERNARY = 3


class GpuPolyError(RuntimeError):
    """Raised where the reference's Rust wrapper panics (`check_status`, gpu.rs:259-263)."""


class GpuRngSeed(C.Structure):
    """`struct { uint64_t words[4]; }` passed by value (gpu.rs:45-61)."""

    _fields_ = [("words", C.c_uint64 * 4)]

    @classmethod
    def from_bytes(cls, b: bytes) -> "GpuRngSeed":
        if len(b) != 32:
            raise ValueError("seed must be 32 bytes")
        s = cls()
        for i in range(4):
            s.words[i] = int.from_bytes(b[8 * i : 8 * i + 8], "little")
        return s

    def to_bytes(self) -> bytes:
        return b"".join(int(w).to_bytes(8, "little") for w in self.words)


class GpuBatchOp(C.Structure):
    """`struct GpuBatchOp { int kind; GpuMatrix *out; const GpuMatrix *lhs, *rhs; }` (include/gpupoly.h)."""

    _fields_ = [("kind", C.c_int), ("out", C.c_void_p), ("lhs", C.c_void_p), ("rhs", C.c_void_p)]


GPUPOLY_OP_MUL, GPUPOLY_OP_ADD, GPUPOLY_OP_SUB, GPUPOLY_OP_MUL_SCALAR, GPUPOLY_OP_NEG, GPUPOLY_OP_DECOMPOSE, GPUPOLY_OP_MUL_DECOMPOSE = range(7)

_vp = C.c_void_p
_sz = C.c_size_t
_u8p = C.POINTER(C.c_uint8)

# name -> (restype, argtypes); every symbol include/gpupoly.h declares
SIGNATURES = {
    "gpu_context_create": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), _sz, C.POINTER(C.c_int), _sz, C.POINTER(_vp)]),
    "gpu_context_destroy": (None, [_vp]),
    "gpu_context_get_N": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gpu_event_set_wait": (C.c_int, [_vp]),
    "gpu_event_set_destroy": (None, [_vp]),
    "gpu_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "gpu_device_mem_info": (C.c_int, [C.c_int, C.POINTER(_sz), C.POINTER(_sz)]),
    "gpu_device_synchronize": (C.c_int, []),
    "gpu_device_reset": (C.c_int, []),
    "gpu_last_error": (C.c_char_p, []),
    "gpu_set_last_error": (C.c_int, [C.c_char_p]),
    "gpu_pinned_alloc": (_vp, [_sz]),
    "gpu_pinned_free": (None, [_vp]),
    "gpu_matrix_create": (C.c_int, [_vp, C.c_int, _sz, _sz, C.c_int, C.POINTER(_vp)]),
    "gpu_matrix_destroy": (None, [_vp]),
    "gpu_matrix_copy": (C.c_int, [_vp, _vp]),
    "gpu_matrix_copy_block": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz]),
    "gpu_matrix_add": (C.c_int, [_vp, _vp, _vp]),
    "gpu_matrix_sub": (C.c_int, [_vp, _vp, _vp]),
    "gpu_matrix_add_block": (C.c_int, [_vp, _vp, _sz, _sz, _sz, _sz, _sz, _sz]),
    "gpu_matrix_mul": (C.c_int, [_vp, _vp, _vp]),
    "gpu_matrix_mul_scalar": (C.c_int, [_vp, _vp, _vp]),
    "gpu_matrix_equal": (C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    "gpu_matrix_ntt_all": (C.c_int, [_vp]),
    "gpu_matrix_intt_all": (C.c_int, [_vp]),
    "gpu_matrix_fill_gadget": (C.c_int, [_vp, C.c_uint32]),
    "gpu_matrix_fill_small_gadget": (C.c_int, [_vp, C.c_uint32]),
    "gpu_matrix_fill_small_decomposed_identity_chunk": (C.c_int, [_vp, _vp, _sz]),
    "gpu_matrix_decompose_base": (C.c_int, [_vp, C.c_uint32, _vp]),
    "gpu_matrix_decompose_base_small": (C.c_int, [_vp, C.c_uint32, _vp]),
    "gpu_matrix_sample_distribution": (C.c_int, [_vp, C.c_int, C.c_double, GpuRngSeed]),
    "gpu_matrix_sample_distribution_columns": (C.c_int, [_vp, C.c_int, C.c_double, GpuRngSeed, _sz, _sz]),
    "gpu_matrix_gauss_samp_gq_arb_base": (C.c_int, [_vp, C.c_uint32, C.c_double, C.c_double, GpuRngSeed, _vp]),
    "gpu_matrix_sample_p1_full": (C.c_int, [_vp, _vp, _vp, _vp, C.c_double, C.c_double, C.c_double, GpuRngSeed, _vp]),
    "gpu_matrix_create_p1_covariance_cache": (C.c_int, [_vp, _vp, _vp, C.c_double, C.c_double, C.c_double, C.POINTER(_vp)]),
    "gpu_matrix_destroy_p1_covariance_cache": (None, [_vp]),
    "gpu_matrix_sample_p1_full_cached": (C.c_int, [_vp, _vp, GpuRngSeed, _vp]),
    "gpu_matrix_load_rns_batch": (C.c_int, [_vp, _vp, _sz, C.c_int, C.POINTER(_vp)]),
    "gpu_matrix_store_rns_batch": (C.c_int, [_vp, _vp, _sz, C.c_int, C.POINTER(_vp)]),
    "gpu_matrix_store_const_coeff_batch": (C.c_int, [_vp, _vp, _sz, C.POINTER(_vp)]),
    "gpu_matrix_store_compact_bytes": (C.c_int, [_vp, _vp, _sz, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(_sz)]),
    "gpu_matrix_load_compact_bytes": (C.c_int, [_vp, _vp, _sz, C.c_uint16]),
    "gpu_poly_store_compact_bytes": (C.c_int, [_vp, _vp, _sz, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.POINTER(_sz)]),
    "gpu_poly_load_compact_bytes": (C.c_int, [_vp, _vp, _sz, C.c_uint16]),
    "gpupoly_matrix_mul_decompose": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "gpupoly_matrix_mul_batch": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_size_t]),
    "gpupoly_batch": (C.c_int, [C.POINTER(GpuBatchOp), C.c_size_t, C.c_uint32]),
    "gpupoly_matrix_mul_decompose_small": (C.c_int, [_vp, _vp, _vp, C.c_uint32]),
    "gpupoly_matrix_mul_tensor_identity": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "gpupoly_matrix_mul_tensor_identity_decompose": (C.c_int, [_vp, _vp, _vp, C.c_size_t, C.c_uint32]),
    "gpupoly_matrix_mul_scalar_intt": (C.c_int, [_vp, _vp, _vp]),
    "gpupoly_matrix_transpose": (C.c_int, [_vp, _vp]),
    "gpupoly_matrix_tensor": (C.c_int, [_vp, _vp, _vp]),
    "gpupoly_matrix_add_rows": (C.c_int, [_vp, C.c_size_t, _vp, _vp]),
    "gpupoly_matrix_ntt_add_rows": (C.c_int, [_vp, C.c_size_t, _vp, _vp, C.c_int]),
    "gpupoly_matrix_row_view": (C.c_int, [_vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gpupoly_matrix_neg": (C.c_int, [_vp, _vp]),
    "gpupoly_matrix_fill_zero": (C.c_int, [_vp]),
    "gpupoly_matrix_fill_identity": (C.c_int, [_vp, _vp]),
    "gpupoly_matrix_sample_decomposed": (C.c_int, [_vp, C.c_int, C.c_double, GpuRngSeed, C.c_uint32, C.c_int]),
    "gpupoly_timer_start": (C.c_int, [_vp]),
    "gpupoly_timer_stop": (C.c_int, [_vp, C.POINTER(C.c_float)]),
    "gpupoly_timer_mark": (C.c_int, [_vp, C.c_uint32]),
    "gpupoly_timer_elapsed": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]),
    "gpupoly_matrix_device_ptr": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_sz)]),
    "gpupoly_matrix_copy_to_context": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "gpupoly_context_device": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gpupoly_context_word_bytes": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gpupoly_context_last_kernel": (C.c_char_p, [_vp]),
    "gpupoly_context_stream": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "gpupoly_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(_vp)]),
    "gpupoly_comm_destroy": (None, [_vp]),
    "gpupoly_comm_size": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "gpupoly_comm_backend": (C.c_char_p, [_vp]),
    "gpupoly_matrix_all_gather_columns": (C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "gpupoly_matrix_sample_distribution_segments": (C.c_int, [_vp, C.c_int, C.c_double, C.POINTER(GpuRngSeed), C.POINTER(_sz), _sz]),
    "gpupoly_matrix_sample_p1_full_cached_segments": (C.c_int, [_vp, _vp, C.POINTER(GpuRngSeed), C.POINTER(_sz), _sz, _vp]),
    "gpupoly_matrix_gauss_samp_gq_arb_base_segments": (C.c_int, [_vp, C.c_uint32, C.c_double, C.c_double, C.POINTER(GpuRngSeed), C.POINTER(_sz), _sz, _vp]),
    "gpupoly_matrix_concat_columns": (C.c_int, [_vp, C.POINTER(C.c_void_p), _sz]),
    "gpupoly_matrix_split_columns": (C.c_int, [_vp, C.POINTER(C.c_void_p), _sz]),
    "gpupoly_launch_count": (C.c_uint64, []),
    "gpupoly_detmath_eval": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), _sz]),
    "gpupoly_device_can_access_peer": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "gpupoly_marker_launch": (C.c_int, [_vp, C.c_uint32]),
    "gpupoly_trace_begin": (C.c_int, []),
    "gpupoly_trace_end": (C.c_char_p, []),
    "gpupoly_version": (C.c_char_p, []),
    "gpupoly_reload_env": (C.c_int, []),
}

_lib = None


def lib() -> C.CDLL:
    """Load libgpupoly.so; fail loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GpuPolyError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C mxx_amd/csrc`). mxx_amd has no CPU fallback."
            )
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error_string() -> str:
    msg = lib().gpu_last_error()
    return msg.decode("utf-8", "replace") if msg else "unknown GPU error"


def check_status(code: int, context: str) -> None:
    if code != 0:
        raise GpuPolyError(f"{context} failed: {last_error_string()}")


def reload_env() -> None:
    """Re-read the MXX_HIP_* switches (libgpupoly caches them per context at creation)."""
    check_status(lib().gpupoly_reload_env(), "gpupoly_reload_env")


def trace_begin() -> None:
    """Start the library's launch trace (every kernel / device copy bracketed by hipEvents on its own stream)."""
    check_status(lib().gpupoly_trace_begin(), "gpupoly_trace_begin")


def trace_end() -> list[dict]:
    """Stop the trace; one dict per launch in launch order: kernel, blocks, threads, bytes (0 = not stated), ms."""
    text = lib().gpupoly_trace_end()
    if text is None:
        raise GpuPolyError(f"gpupoly_trace_end failed: {last_error_string()}")
    out = []
    for ln in text.decode().splitlines():
        name, blocks, threads, nbytes, ms = ln.split("\t")
        out.append({"kernel": name.strip("()"), "blocks": int(blocks), "threads": int(threads), "bytes": float(nbytes), "ms": float(ms)})
    return out


def gpu_device_sync() -> None:
    check_status(lib().gpu_device_synchronize(), "gpu_device_synchronize")


def detected_gpu_device_ids() -> list[int]:
    n = C.c_int(0)
    if lib().gpu_device_count(C.byref(n)) != 0 or n.value <= 0:
        return []
    return list(range(n.value))


def peer_access_matrix(devices=None) -> list[list[int]]:
    """[i][j] = 1 when device i can address device j's memory directly (xGMI peer mapping)."""
    devs = detected_gpu_device_ids() if devices is None else list(devices)
    out = []
    for a in devs:
        row = []
        for b in devs:
            can = C.c_int(0)
            check_status(lib().gpupoly_device_can_access_peer(a, b, C.byref(can)), "gpupoly_device_can_access_peer")
            row.append(int(can.value))
        out.append(row)
    return out


def detected_gpu_device_count() -> int:
    return len(detected_gpu_device_ids())


def wait_and_destroy_events(events: _vp) -> None:
    if events:
        st = lib().gpu_event_set_wait(events)
        lib().gpu_event_set_destroy(events)
        check_status(st, "gpu_event_set_wait")
