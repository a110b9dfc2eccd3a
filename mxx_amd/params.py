"""Ring parameters and device contexts.

Mirrors, for the GPU path only, the reference's
  * `DCRTPolyParams`     — src/poly/dcrt/params.rs:1-110
  * `GpuDCRTPolyParams`  — src/poly/dcrt/gpu.rs:422-650
  * `GpuContext`         — src/poly/dcrt/gpu.rs:652-705
(`PolyParams` trait: src/poly/mod.rs:18-77).
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref
from dataclasses import dataclass, field

from . import _ffi


# ---------------------------------------------------------------------------
# CRT basis — OpenFHE ILDCRTParams(order=2n, depth, bits): LastPrime / PreviousPrime
# (called by the reference through ffi::GenCRTBasis, params.rs:60-66)
# ---------------------------------------------------------------------------
def _is_prime(n: int) -> bool:
    if n < 2:
        return False
    small = (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37)
    for p in small:
        if n % p == 0:
            return n == p
    d, s = n - 1, 0
    while d % 2 == 0:
        d //= 2
        s += 1
    for a in small:
        x = pow(a, d, n)
        if x in (1, n - 1):
            continue
        for _ in range(s - 1):
            x = x * x % n
            if x == n - 1:
                break
        else:
            return False
    return True


def gen_crt_basis(ring_dimension: int, crt_depth: int, crt_bits: int) -> list[int]:
    m = 2 * ring_dimension
    q = (1 << crt_bits) + 1
    out = []
    for _ in range(crt_depth):
        while True:
            q -= m
            if q <= m:
                raise ValueError("ran out of primes")
            if _is_prime(q):
                break
        if q.bit_length() != crt_bits:
            raise ValueError(f"not enough {crt_bits}-bit primes = 1 mod {m}")
        out.append(q)
    return out


class DCRTPolyParams:
    """`DCRTPolyParams::new(ring_dimension, crt_depth, crt_bits, base_bits)` (params.rs:77-98)."""

    def __init__(self, ring_dimension: int = 4, crt_depth: int = 2, crt_bits: int = 17, base_bits: int = 1):
        if ring_dimension & (ring_dimension - 1):
            raise ValueError("ring_dimension must be a power of 2")
        self._n = ring_dimension
        self._depth = crt_depth
        self._crt_bits = crt_bits
        self._base_bits = base_bits
        self._moduli = gen_crt_basis(ring_dimension, crt_depth, crt_bits)
        self._modulus = 1
        for q in self._moduli:
            self._modulus *= q
        if crt_bits % base_bits == 0:
            self._last_mask = None
        else:
            dpt = -(-crt_bits // base_bits)
            self._last_mask = (1 << (crt_bits - base_bits * (dpt - 1))) - 1

    def ring_dimension(self) -> int:
        return self._n

    def modulus(self) -> int:
        return self._modulus

    def base_bits(self) -> int:
        return self._base_bits

    def modulus_bits(self) -> int:
        return self._modulus.bit_length()

    def modulus_digits(self) -> int:
        return -(-self._crt_bits // self._base_bits) * self._depth

    def crt_depth(self) -> int:
        return self._depth

    def crt_bits(self) -> int:
        return self._crt_bits

    def decompose_last_mask(self):
        return self._last_mask

    def to_crt(self):
        return list(self._moduli), self._crt_bits, self._depth

    def __eq__(self, other):
        return isinstance(other, DCRTPolyParams) and (self._n, self._moduli, self._base_bits) == (
            other._n,
            other._moduli,
            other._base_bits,
        )

    def __hash__(self):
        return hash((self._n, tuple(self._moduli), self._base_bits))


# ---------------------------------------------------------------------------
# device context
# ---------------------------------------------------------------------------
class GpuContext:
    """Owning handle of a `GpuContext*` (gpu.rs:652-705)."""

    def __init__(self, log_n: int, moduli: list[int], gpu_ids: list[int], dnum: int):
        lib = _ffi.lib()
        arr = (C.c_uint64 * len(moduli))(*moduli)
        ids = (C.c_int * len(gpu_ids))(*gpu_ids)
        raw = C.c_void_p()
        st = lib.gpu_context_create(log_n, len(moduli) - 1, dnum, arr, len(moduli), ids, len(gpu_ids), C.byref(raw))
        _ffi.check_status(st, "gpu_context_create")
        self.raw = raw
        self.gpu_ids = list(gpu_ids)
        self._finalizer = weakref.finalize(self, lib.gpu_context_destroy, raw)

    def device(self) -> int:
        d = C.c_int(0)
        _ffi.check_status(_ffi.lib().gpupoly_context_device(self.raw, C.byref(d)), "gpupoly_context_device")
        return d.value

    def stream_handle(self) -> int:
        """The context's compute stream as an integer hipStream_t (for `torch.cuda.ExternalStream`)."""
        h = C.c_void_p()
        _ffi.check_status(_ffi.lib().gpupoly_context_stream(self.raw, C.byref(h)), "gpupoly_context_stream")
        return h.value or 0

    def last_kernel(self) -> str:
        """Name of the product kernel the dispatcher launched last on this context."""
        return (_ffi.lib().gpupoly_context_last_kernel(self.raw) or b"").decode()

    def word_bytes(self) -> int:
        d = C.c_int(0)
        _ffi.check_status(_ffi.lib().gpupoly_context_word_bytes(self.raw, C.byref(d)), "gpupoly_context_word_bytes")
        return d.value

    def marker(self, ident: int) -> None:
        """A one-thread no-op kernel on the compute stream (delimits a timed region in a profiler's dispatch list)."""
        _ffi.check_status(_ffi.lib().gpupoly_marker_launch(self.raw, ident), "gpupoly_marker_launch")

    def timer_start(self) -> None:
        _ffi.check_status(_ffi.lib().gpupoly_timer_start(self.raw), "gpupoly_timer_start")

    def timer_mark(self, slot: int) -> None:
        _ffi.check_status(_ffi.lib().gpupoly_timer_mark(self.raw, slot), "gpupoly_timer_mark")

    def timer_elapsed(self, slot_begin: int, slot_end: int) -> float:
        ms = C.c_float(0)
        st = _ffi.lib().gpupoly_timer_elapsed(self.raw, slot_begin, slot_end, C.byref(ms))
        _ffi.check_status(st, "gpupoly_timer_elapsed")
        return float(ms.value)

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        _ffi.check_status(_ffi.lib().gpupoly_timer_stop(self.raw, C.byref(ms)), "gpupoly_timer_stop")
        return float(ms.value)


_ctx_cache: "dict[tuple, weakref.ReferenceType[GpuContext]]" = {}
_ctx_cache_lock = threading.Lock()


def _cached_context(n: int, moduli: list[int], base_bits: int, gpu_ids: list[int], dnum: int) -> GpuContext:
    """Process-wide per-device context cache (gpu.rs:430-449,531-557)."""
    key = (n, tuple(moduli), base_bits, tuple(gpu_ids), dnum)
    with _ctx_cache_lock:
        ref = _ctx_cache.get(key)
        ctx = ref() if ref is not None else None
        if ctx is None:
            ctx = GpuContext(n.bit_length() - 1, moduli, gpu_ids, dnum)
            _ctx_cache[key] = weakref.ref(ctx)
        return ctx


class GpuDCRTPolyParams:
    """`GpuDCRTPolyParams::new(ring_dimension, moduli, base_bits)` (gpu.rs:559-594).

    Like the reference, the default takes only the FIRST detected GPU; callers fan
    work out with `device_ids()` / `params_for_device(id)` (src/poly/mod.rs:36-43).
    """

    def __init__(self, ring_dimension: int, moduli: list[int], base_bits: int, gpu_ids=None, dnum=None):
        if ring_dimension & (ring_dimension - 1) or ring_dimension < 2:
            raise ValueError("ring_dimension must be a power of 2")
        if not moduli:
            raise ValueError("moduli must not be empty")
        if gpu_ids is None:
            detected = _ffi.detected_gpu_device_ids()
            if not detected:
                raise _ffi.GpuPolyError("no GPU device detected")
            gpu_ids = detected[:1]
        self._n = ring_dimension
        self._moduli = [int(q) for q in moduli]
        self._base_bits = int(base_bits)
        self._crt_bits = max(q.bit_length() for q in self._moduli)
        self._gpu_ids = list(gpu_ids)
        self._dnum = int(dnum) if dnum is not None else len(self._gpu_ids)
        self._modulus = 1
        for q in self._moduli:
            self._modulus *= q
        self._ctx = _cached_context(self._n, self._moduli, self._base_bits, self._gpu_ids, self._dnum)

    @classmethod
    def new(cls, ring_dimension: int, moduli, base_bits: int) -> "GpuDCRTPolyParams":
        return cls(ring_dimension, moduli, base_bits)

    @classmethod
    def new_with_gpu(cls, ring_dimension: int, moduli, base_bits: int, gpu_ids, dnum=None) -> "GpuDCRTPolyParams":
        """gpu.rs:567-594: explicit device list; dnum defaults to the number of devices (1 for an empty list)."""
        gpu_ids = list(gpu_ids)
        return cls(ring_dimension, moduli, base_bits, gpu_ids=gpu_ids, dnum=dnum if dnum is not None else max(len(gpu_ids), 1))

    @classmethod
    def from_cpu_params(cls, params: DCRTPolyParams, gpu_ids=None) -> "GpuDCRTPolyParams":
        moduli, _, _ = params.to_crt()
        return cls(params.ring_dimension(), moduli, params.base_bits(), gpu_ids=gpu_ids)

    # PolyParams trait
    def ring_dimension(self) -> int:
        return self._n

    def modulus(self) -> int:
        return self._modulus

    def base_bits(self) -> int:
        return self._base_bits

    def modulus_bits(self) -> int:
        return self._modulus.bit_length()

    def modulus_digits(self) -> int:
        return -(-self._crt_bits // self._base_bits) * len(self._moduli)

    def to_crt(self):
        return list(self._moduli), self._crt_bits, len(self._moduli)

    def device_ids(self) -> list[int]:
        return _ffi.detected_gpu_device_ids()

    def params_for_device(self, device_id: int) -> "GpuDCRTPolyParams":
        return GpuDCRTPolyParams(self._n, self._moduli, self._base_bits, gpu_ids=[device_id], dnum=1)

    # inherent
    def crt_depth(self) -> int:
        return len(self._moduli)

    def crt_bits(self) -> int:
        return self._crt_bits

    def moduli(self) -> list[int]:
        return list(self._moduli)

    def gpu_ids(self) -> list[int]:
        return list(self._gpu_ids)

    def ctx(self) -> GpuContext:
        return self._ctx

    def ctx_raw(self):
        return self._ctx.raw

    def modulus_for_level(self, level: int) -> int:
        out = 1
        for q in self._moduli[: level + 1]:
            out *= q
        return out

    def reconstruct_coeffs_for_level(self, level: int) -> list:
        """CRT weights (Q / q_i) * ((Q / q_i)^-1 mod q_i) mod Q for the limbs 0..=level (gpu.rs:620-635)."""
        Q = self.modulus_for_level(level)
        out = []
        for q in self._moduli[: level + 1]:
            Qi = Q // q
            out.append(Qi * pow(Qi % q, -1, q) % Q)
        return out

    def __eq__(self, other):
        return (
            isinstance(other, GpuDCRTPolyParams)
            and self._n == other._n
            and self._moduli == other._moduli
            and self._base_bits == other._base_bits
            and self._gpu_ids == other._gpu_ids
        )

    def __hash__(self):
        return hash((self._n, tuple(self._moduli), self._base_bits, tuple(self._gpu_ids)))

    def __repr__(self):
        return (
            f"GpuDCRTPolyParams(n={self._n}, crt_depth={len(self._moduli)}, crt_bits={self._crt_bits}, "
            f"base_bits={self._base_bits}, gpu_ids={self._gpu_ids})"
        )
