"""`GpuDCRTPolyMatrix` — host-side mirror of the reference's GPU matrix wrapper.

Same names, argument meaning and error behaviour as
`src/matrix/gpu_dcrt_poly.rs:221-1678,1716-1895` (the `PolyMatrix` trait is
`src/matrix/mod.rs:45-379`); every method is a thin sequence of C-ABI calls into
libgpupoly (include/gpupoly.h).  Host data crosses the boundary as numpy uint64
arrays in the wire layout (rows, cols, level+1, n) — the `[poly][limb][n]` u64
layout of `load/store_rns_bytes` (gpu_dcrt_poly.rs:576-663).
"""
from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import _ffi
from ._ffi import GPU_POLY_FORMAT_COEFF, GPU_POLY_FORMAT_EVAL, GpuRngSeed, check_status
from .params import GpuDCRTPolyParams


def mul_decompose_column_chunk_width_is_set() -> bool:
    return bool(os.environ.get("MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH"))


def mul_decompose_column_chunk_width() -> int:
    """`MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH`, default 1 (src/env.rs of the reference)."""
    try:
        return max(1, int(os.environ.get("MXX_MUL_DECOMPOSE_COLUMN_CHUNK_WIDTH", "1")))
    except ValueError:
        return 1


def _bincode_varint(v: int) -> bytes:
    """bincode 2 `config::standard()` unsigned varint (u16/u32/u64/usize)."""
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return bytes([251]) + v.to_bytes(2, "little")
    if v < 1 << 32:
        return bytes([252]) + v.to_bytes(4, "little")
    return bytes([253]) + v.to_bytes(8, "little")


def _bincode_read_varint(data: bytes, pos: int):
    tag = data[pos]
    if tag < 251:
        return tag, pos + 1
    width = {251: 2, 252: 4, 253: 8}[tag]
    return int.from_bytes(data[pos + 1 : pos + 1 + width], "little"), pos + 1 + width


def block_size() -> int:
    """`BLOCK_SIZE`, default 100 (src/env.rs:175-178): entries per side of a stored matrix block."""
    try:
        value = int(os.environ.get("BLOCK_SIZE", "100"))
    except ValueError:
        return 100
    return value if value >= 1 else 100  # the reference parses a usize: "0" would never advance, a negative value does not parse


def block_offsets(rng: range, block: int) -> list:
    """[start, start + block, ..., stop] (gpu_dcrt_poly.rs:1899-1909)."""
    assert block > 0, "block_offsets: block must be positive"
    offsets, cur = [rng.start], rng.start
    while cur < rng.stop:
        cur = min(cur + block, rng.stop)
        offsets.append(cur)
    return offsets


def rns_bytes_len_for_level(params, level: int) -> int:
    """(level + 1) * n * 8: one polynomial in the `[limb][n]` u64 wire layout (gpu_dcrt_poly.rs:1916-1920)."""
    assert level < params.crt_depth(), "invalid RNS byte length level"
    return (level + 1) * params.ring_dimension() * 8


def rns_bytes_len(params) -> int:
    """gpu_dcrt_poly.rs:1911-1914."""
    return rns_bytes_len_for_level(params, max(params.crt_depth() - 1, 0))


def one_rns_bytes(params) -> bytes:
    """EVAL wire bytes of the constant polynomial 1 (gpu_dcrt_poly.rs:1922-1932)."""
    if rns_bytes_len(params) == 0:
        return b""
    one = GpuDCRTPolyMatrix.identity(params, 1)
    return one.to_rns().tobytes()


class GpuDCRTMatrixRnsSnapshot:
    """Host copy of a matrix in the u64 wire layout with its shape, level and format tag
    (gpu_dcrt_poly.rs:72-120): what callers keep between devices / across a device reset."""

    __slots__ = ("_nrow", "_ncol", "_level", "_is_ntt", "_bytes_per_poly", "_bytes")

    def __init__(self, nrow, ncol, level, is_ntt, bytes_per_poly, data):
        self._nrow, self._ncol, self._level, self._is_ntt = int(nrow), int(ncol), int(level), bool(is_ntt)
        self._bytes_per_poly = int(bytes_per_poly)
        self._bytes = bytes(data)

    def nrow(self) -> int:
        return self._nrow

    def ncol(self) -> int:
        return self._ncol

    def level(self) -> int:
        return self._level

    def is_ntt(self) -> bool:
        return self._is_ntt

    def bytes_per_poly(self) -> int:
        return self._bytes_per_poly

    def bytes(self) -> bytes:
        return self._bytes

    def validate_for_params(self, params) -> None:
        assert self._level < params.crt_depth(), "invalid RNS snapshot level"
        assert self._bytes_per_poly == rns_bytes_len_for_level(params, self._level), "RNS snapshot bytes_per_poly mismatch"
        assert len(self._bytes) == self._nrow * self._ncol * self._bytes_per_poly, "RNS snapshot byte length mismatch"

    def __eq__(self, other) -> bool:
        if not isinstance(other, GpuDCRTMatrixRnsSnapshot):
            return NotImplemented
        return all(getattr(self, f) == getattr(other, f) for f in self.__slots__)

    def __repr__(self) -> str:
        return (f"GpuDCRTMatrixRnsSnapshot(nrow={self._nrow}, ncol={self._ncol}, level={self._level}, "
                f"is_ntt={self._is_ntt}, bytes_per_poly={self._bytes_per_poly}, bytes={len(self._bytes)})")


def _bincode_read_nested_bytes(data: bytes) -> list:
    """bincode 2 `config::standard()` Vec<Vec<Vec<u8>>>: varint lengths, raw bytes innermost."""
    pos = 0
    nrows, pos = _bincode_read_varint(data, pos)
    out = []
    for _ in range(nrows):
        ncols, pos = _bincode_read_varint(data, pos)
        row = []
        for _ in range(ncols):
            blen, pos = _bincode_read_varint(data, pos)
            assert pos + blen <= len(data), "truncated matrix block file"
            row.append(data[pos : pos + blen])
            pos += blen
        out.append(row)
    return out


def _bincode_nested_bytes(entries) -> bytes:
    """inverse of `_bincode_read_nested_bytes` (what the reference's `write_to_files` side produces)."""
    parts = [_bincode_varint(len(entries))]
    for row in entries:
        parts.append(_bincode_varint(len(row)))
        for e in row:
            parts.append(_bincode_varint(len(e)))
            parts.append(bytes(e))
    return b"".join(parts)


# ---- pinned host staging, one grow-only buffer per thread (compact wire format: device -> host at PCIe rate) ----------
_pinned_tls = threading.local()


def _pinned_capacity() -> int:
    return getattr(_pinned_tls, "size", 0)


def _pinned_buffer(nbytes: int) -> int:
    """address of this thread's pinned buffer, at least `nbytes` long (gpu_pinned_alloc; released when the thread ends)"""
    if _pinned_capacity() < nbytes:
        old = getattr(_pinned_tls, "holder", None)
        if old is not None:
            old.release()
        size = max(nbytes, 1 << 20)
        ptr = _ffi.lib().gpu_pinned_alloc(size)
        if not ptr:
            _pinned_tls.holder, _pinned_tls.size = None, 0
            raise _ffi.GpuPolyError(f"gpu_pinned_alloc({size}) failed: {_ffi.last_error_string()}")
        _pinned_tls.holder, _pinned_tls.size = _PinnedBlock(ptr), size
    return _pinned_tls.holder.ptr


class _PinnedBlock:
    def __init__(self, ptr):
        self.ptr = ptr
        self._fin = weakref.finalize(self, _ffi.lib().gpu_pinned_free, C.c_void_p(ptr))

    def release(self):
        self._fin()


class GpuP1CovarianceCache:
    def __init__(self, raw, params=None):
        self.raw = raw
        self.params = params  # the C object frees into its context's allocator: the context must outlive it (the Rust side holds an Arc)
        self._finalizer = weakref.finalize(self, _ffi.lib().gpu_matrix_destroy_p1_covariance_cache, raw)


class GpuDCRTPolyMatrix:
    __slots__ = ("params", "nrow", "ncol", "level", "is_ntt", "raw", "_finalizer", "_parent", "_version", "__weakref__")

    # ------------------------------------------------------------------ construction
    def __init__(self, params: GpuDCRTPolyParams, nrow: int, ncol: int, level: int, is_ntt: bool):
        """`new_empty_with_state` (gpu_dcrt_poly.rs:222-256): contents undefined."""
        if not (0 <= level < params.crt_depth()):
            raise AssertionError("invalid level for matrix create")
        raw = C.c_void_p()
        fmt = GPU_POLY_FORMAT_EVAL if is_ntt else GPU_POLY_FORMAT_COEFF
        st = _ffi.lib().gpu_matrix_create(params.ctx_raw(), level, nrow, ncol, fmt, C.byref(raw))
        check_status(st, f"gpu_matrix_create(nrow={nrow}, ncol={ncol}, level={level}, format={fmt})")
        self.params = params
        self.nrow = nrow
        self.ncol = ncol
        self.level = level
        self.is_ntt = is_ntt
        self.raw = raw
        self._parent = None  # set on row views: the matrix whose storage this one shares
        self._version = 0  # bumped by every in-place write (and on the parent of a view): keys caches derived from the contents
        self._finalizer = weakref.finalize(self, _ffi.lib().gpu_matrix_destroy, raw)

    @classmethod
    def new_empty(cls, params, nrow, ncol) -> "GpuDCRTPolyMatrix":
        return cls(params, nrow, ncol, params.crt_depth() - 1, True)

    @classmethod
    def zero(cls, params, nrow, ncol) -> "GpuDCRTPolyMatrix":
        return cls._new_zero_with_state(params, nrow, ncol, params.crt_depth() - 1, True)

    new_zero = zero  # gpu_dcrt_poly.rs:367-370

    @classmethod
    def _new_zero_with_state(cls, params, nrow, ncol, level, is_ntt) -> "GpuDCRTPolyMatrix":
        out = cls(params, nrow, ncol, level, is_ntt)
        if nrow == 0 or ncol == 0:
            return out
        # the reference uploads a host vector of zeros (gpu_dcrt_poly.rs:343-365); one device memset here
        check_status(_ffi.lib().gpupoly_matrix_fill_zero(out.raw), "gpupoly_matrix_fill_zero")
        return out

    @classmethod
    def from_rns(cls, params, data: np.ndarray, eval_format: bool) -> "GpuDCRTPolyMatrix":
        """Upload wire-layout residues (rows, cols, L, n)."""
        data = np.ascontiguousarray(data, dtype=np.uint64)
        rows, cols, L, n = data.shape
        if n != params.ring_dimension():
            raise ValueError("ring dimension mismatch")
        out = cls(params, rows, cols, L - 1, eval_format)
        out.load_rns(data, eval_format)
        return out

    @classmethod
    def from_cpu_matrix(cls, params, coeff_residues: np.ndarray) -> "GpuDCRTPolyMatrix":
        """`from_cpu_matrix` (gpu_dcrt_poly.rs:769-817): CPU matrices travel as EVAL residues."""
        return cls.from_rns(params, coeff_residues, True)

    @classmethod
    def identity(cls, params, size, scalar=None) -> "GpuDCRTPolyMatrix":
        """`identity` (gpu_dcrt_poly.rs:1158-1188); built on the device (gpupoly_matrix_fill_identity)."""
        out = cls.new_empty(params, size, size)
        if size == 0:
            return out
        s_raw = None
        if scalar is not None:
            sm = scalar.inner if hasattr(scalar, "inner") else scalar
            sm = sm.ensure_eval()
            s_raw = sm.raw
        check_status(_ffi.lib().gpupoly_matrix_fill_identity(out.raw, s_raw), "gpupoly_matrix_fill_identity")
        return out

    @classmethod
    def gadget_matrix(cls, params, size) -> "GpuDCRTPolyMatrix":
        if size == 0:
            return cls.zero(params, 0, 0)
        out = cls.new_empty(params, size, size * params.modulus_digits())
        check_status(_ffi.lib().gpu_matrix_fill_gadget(out.raw, params.base_bits()), "gpu_matrix_fill_gadget")
        out.is_ntt = True
        return out

    @classmethod
    def small_gadget_matrix(cls, params, size) -> "GpuDCRTPolyMatrix":
        if size == 0:
            return cls.zero(params, 0, 0)
        k = -(-params.crt_bits() // params.base_bits())
        out = cls.new_empty(params, size, size * k)
        check_status(_ffi.lib().gpu_matrix_fill_small_gadget(out.raw, params.base_bits()), "gpu_matrix_fill_small_gadget")
        out.is_ntt = True
        return out

    # ------------------------------------------------------------------ host transfer
    def _bytes_per_poly(self) -> int:
        return (self.level + 1) * self.params.ring_dimension() * 8

    def _touch(self) -> None:
        """contents (or the format tag) are about to change in place: anything cached from them is stale"""
        m = self
        while m is not None:
            m._version += 1
            m = m._parent

    def content_version(self) -> int:
        """changes whenever this object's storage was rewritten in place through the host mirror (own writes and writes
        through row views); caches derived from the contents key on (object, content_version())"""
        return self._version

    def load_rns(self, data: np.ndarray, eval_format: bool) -> None:
        """`load_rns_bytes` (gpu_dcrt_poly.rs:629-663)."""
        data = np.ascontiguousarray(data, dtype=np.uint64)
        if data.size == 0:
            self.is_ntt = eval_format
            return
        if data.size != self.nrow * self.ncol * (self.level + 1) * self.params.ring_dimension():
            raise ValueError("load_rns: size mismatch")
        events = C.c_void_p()
        fmt = GPU_POLY_FORMAT_EVAL if eval_format else GPU_POLY_FORMAT_COEFF
        self._touch()
        st = _ffi.lib().gpu_matrix_load_rns_batch(self.raw, data.ctypes.data, self._bytes_per_poly(), fmt, C.byref(events))
        check_status(st, "gpu_matrix_load_rns_batch")
        _ffi.wait_and_destroy_events(events)
        self.is_ntt = eval_format

    def to_rns(self) -> np.ndarray:
        """`store_rns_bytes` in the current format (gpu_dcrt_poly.rs:576-596)."""
        n = self.params.ring_dimension()
        out = np.empty((self.nrow, self.ncol, self.level + 1, n), dtype=np.uint64)  # every word is written below
        if out.size == 0:
            return out
        events = C.c_void_p()
        fmt = GPU_POLY_FORMAT_EVAL if self.is_ntt else GPU_POLY_FORMAT_COEFF
        st = _ffi.lib().gpu_matrix_store_rns_batch(self.raw, out.ctypes.data, self._bytes_per_poly(), fmt, C.byref(events))
        check_status(st, "gpu_matrix_store_rns_batch")
        _ffi.wait_and_destroy_events(events)
        return out

    # ---- the reference's byte-slice forms of the same transfers (gpu_dcrt_poly.rs:576-596,629-711) ----------------
    def bytes_per_poly(self) -> int:
        return self._bytes_per_poly()

    def load_rns_bytes(self, data, bytes_per_poly: int, fmt: int) -> None:
        """`load_rns_bytes(bytes, bytes_per_poly, format)`: polynomials `bytes_per_poly` apart, retagged to `fmt`."""
        if len(data) == 0 or bytes_per_poly == 0:
            return
        buf = np.frombuffer(data, dtype=np.uint8)
        assert len(buf) >= self.nrow * self.ncol * bytes_per_poly, "load_rns_bytes: buffer too small"
        events = C.c_void_p()
        self._touch()
        st = _ffi.lib().gpu_matrix_load_rns_batch(self.raw, buf.ctypes.data, bytes_per_poly, fmt, C.byref(events))
        check_status(
            st,
            f"gpu_matrix_load_rns_batch(nrow={self.nrow}, ncol={self.ncol}, level={self.level}, current_ntt={self.is_ntt}, "
            f"format={fmt}, bytes={len(buf)}, bytes_per_poly={bytes_per_poly}, ring_dim={self.params.ring_dimension()}, "
            f"crt_depth={self.params.crt_depth()})",
        )
        _ffi.wait_and_destroy_events(events)
        self.is_ntt = fmt == GPU_POLY_FORMAT_EVAL

    def store_rns_bytes(self, bytes_out, bytes_per_poly: int, fmt: int) -> None:
        """`store_rns_bytes(bytes_out, bytes_per_poly, format)` into a writable buffer (bytearray / numpy uint8);
        `fmt` must be the matrix's current format (the library refuses a conversion, MatrixSerde.cu:765-768)."""
        if len(bytes_out) == 0 or bytes_per_poly == 0:
            return
        buf = np.frombuffer(bytes_out, dtype=np.uint8)
        assert buf.flags.writeable, "store_rns_bytes needs a writable buffer"
        assert len(buf) >= self.nrow * self.ncol * bytes_per_poly, "store_rns_bytes: buffer too small"
        events = C.c_void_p()
        st = _ffi.lib().gpu_matrix_store_rns_batch(self.raw, buf.ctypes.data, bytes_per_poly, fmt, C.byref(events))
        check_status(st, "gpu_matrix_store_rns_batch")
        _ffi.wait_and_destroy_events(events)

    def to_rns_snapshot(self) -> "GpuDCRTMatrixRnsSnapshot":
        bpp = rns_bytes_len_for_level(self.params, self.level)
        data = bytearray(self.nrow * self.ncol * bpp)
        self.store_rns_bytes(data, bpp, GPU_POLY_FORMAT_EVAL if self.is_ntt else GPU_POLY_FORMAT_COEFF)
        return GpuDCRTMatrixRnsSnapshot(self.nrow, self.ncol, self.level, self.is_ntt, bpp, data)

    @classmethod
    def from_rns_snapshot(cls, params, snapshot: "GpuDCRTMatrixRnsSnapshot") -> "GpuDCRTPolyMatrix":
        snapshot.validate_for_params(params)
        out = cls(params, snapshot.nrow(), snapshot.ncol(), snapshot.level(), snapshot.is_ntt())
        if snapshot.bytes():
            out.load_rns_bytes(snapshot.bytes(), snapshot.bytes_per_poly(),
                               GPU_POLY_FORMAT_EVAL if snapshot.is_ntt() else GPU_POLY_FORMAT_COEFF)
        return out

    def load_rns_snapshot(self, snapshot: "GpuDCRTMatrixRnsSnapshot") -> None:
        snapshot.validate_for_params(self.params)
        assert self.nrow == snapshot.nrow(), "RNS snapshot row count mismatch"
        assert self.ncol == snapshot.ncol(), "RNS snapshot column count mismatch"
        assert self.level == snapshot.level(), "RNS snapshot level mismatch"
        assert self.is_ntt == snapshot.is_ntt(), "RNS snapshot format mismatch"
        if not snapshot.bytes():
            return
        self.load_rns_bytes(snapshot.bytes(), snapshot.bytes_per_poly(),
                            GPU_POLY_FORMAT_EVAL if snapshot.is_ntt() else GPU_POLY_FORMAT_COEFF)

    def to_cpu_matrix(self) -> np.ndarray:
        """`to_cpu_matrix` (gpu_dcrt_poly.rs:722-767): the CPU side receives EVAL residues, (rows, cols, L, n) u64."""
        return self.to_eval_rns()

    @classmethod
    def read_from_files(cls, params, nrow: int, ncol: int, dir_path, ident: str) -> "GpuDCRTPolyMatrix":
        """`read_from_files` (gpu_dcrt_poly.rs:1594-1641): blocks of `BLOCK_SIZE` x `BLOCK_SIZE` entries, one file
        `{id}_{bsize}_{r0}.{r1}_{c0}.{c1}.matrix` each, holding bincode(Vec<Vec<Vec<u8>>>) of EVAL wire bytes per entry;
        short or missing entries are zero-padded, as there."""
        bsize = min(block_size(), max(nrow, 1), max(ncol, 1))
        out = cls.new_empty(params, nrow, ncol)
        bpp = rns_bytes_len(params)
        rows_off, cols_off = block_offsets(range(0, nrow), bsize), block_offsets(range(0, ncol), bsize)
        for r0, r1 in zip(rows_off, rows_off[1:]):
            for c0, c1 in zip(cols_off, cols_off[1:]):
                path = os.path.join(os.fspath(dir_path), f"{ident}_{bsize}_{r0}.{r1}_{c0}.{c1}.matrix")
                try:
                    with open(path, "rb") as fh:
                        entries = _bincode_read_nested_bytes(fh.read())
                except OSError as e:
                    raise RuntimeError(f"Failed to read matrix file {path!r}") from e
                rl, cl = r1 - r0, c1 - c0
                flat = bytearray(rl * cl * bpp)
                for i in range(rl):
                    for j in range(cl):
                        if i < len(entries) and j < len(entries[i]):
                            src = entries[i][j][:bpp]
                            start = (i * cl + j) * bpp
                            flat[start : start + len(src)] = src
                block = cls.new_empty(params, rl, cl)
                block.load_rns_bytes(flat, bpp, GPU_POLY_FORMAT_EVAL)
                out.copy_block_from(block, r0, c0, 0, 0, rl, cl)
        return out

    def to_coeff_rns(self) -> np.ndarray:
        return self.ensure_coeff().to_rns()

    def to_eval_rns(self) -> np.ndarray:
        return self.ensure_eval().to_rns()

    def store_const_coeff_words(self) -> np.ndarray:
        """`store_const_coeff_words` (gpu_dcrt_poly.rs:598-627); COEFF format required."""
        L = self.level + 1
        out = np.zeros((self.nrow, self.ncol, L), dtype=np.uint64)
        if out.size == 0:
            return out
        events = C.c_void_p()
        st = _ffi.lib().gpu_matrix_store_const_coeff_batch(self.raw, out.ctypes.data, L, C.byref(events))
        check_status(st, "gpu_matrix_store_const_coeff_batch")
        _ffi.wait_and_destroy_events(events)
        return out

    def coeffs(self) -> list:
        """CRT-reconstructed coefficients as python ints, [row][col][i] (gpu.rs:959-994)."""
        res = self.to_coeff_rns()
        moduli = self.params.moduli()[: self.level + 1]
        Q = 1
        for q in moduli:
            Q *= q
        weights = []
        for q in moduli:
            Qi = Q // q
            weights.append(Qi * pow(Qi, -1, q))
        out = []
        for r in range(self.nrow):
            row = []
            for c in range(self.ncol):
                vals = [0] * res.shape[-1]
                for l, w in enumerate(weights):
                    limb = res[r, c, l]
                    for i in range(len(vals)):
                        vals[i] += int(limb[i]) * w
                row.append([v % Q for v in vals])
            out.append(row)
        return out

    # ------------------------------------------------------------------ compact wire format
    def to_compact_bytes(self) -> bytes:
        """`into_compact_bytes` (gpu_dcrt_poly.rs:956-1002): bincode(standard) tuple
        (1u8, format u8, level u32, nrow usize, ncol usize, max_coeff_bits u16, bytes_per_coeff u16, payload)."""
        return self.clone().into_compact_bytes()

    def into_compact_bytes(self) -> bytes:
        """`into_compact_bytes` (gpu_dcrt_poly.rs:956-1002) as an immutable `bytes`: the framed payload of
        `into_compact_view`, copied once."""
        return bytes(self.into_compact_view())

    def into_compact_view(self) -> memoryview:
        """The same bytes WITHOUT a host copy: a memoryview of this thread's pinned staging buffer, valid until the thread's
        next `into_compact_view` / `into_compact_bytes` call - for a consumer that writes them out at once (a file, a
        socket).  The ABI call copies the payload device -> pinned host memory (PCIe rate; a pageable destination - what
        round 2's mirror handed over - goes through the runtime's bounce buffer and first-touch faults: 15.7 of that
        call's 20.6 ms at the M3A preimage).  The worst-case capacity the ABI wants (the Rust side's `vec![0u8; cap]`) is
        only reserved address space here: pinned memory is grown to the largest payload seen, and a payload that does
        not fit the current buffer is retried once with the exact size the first call reported."""
        fmt = GPU_POLY_FORMAT_EVAL if self.is_ntt else GPU_POLY_FORMAT_COEFF
        coeff_count = self.nrow * self.ncol * self.params.ring_dimension()
        bits_upper = sum(q.bit_length() for q in self.params.moduli()[: self.level + 1])
        cap = (coeff_count * bits_upper + 7) // 8
        head_room = 64  # the bincode header (6 varints) is written in front of the payload afterwards
        max_bits, bpc, plen = C.c_uint16(0), C.c_uint16(0), C.c_size_t(0)
        self._touch()  # an EVAL matrix is taken to the coefficient domain in place
        # first try: whatever this thread already holds (at least 1/8 of the worst case: Gaussian-sized entries need far less)
        want = min(max(cap, 1), max(_pinned_capacity() - head_room, (cap + 7) // 8, 1 << 16))
        for attempt in range(2):
            base = _pinned_buffer(head_room + want)
            st = _ffi.lib().gpu_matrix_store_compact_bytes(
                self.raw, C.c_void_p(base + head_room), want, C.byref(max_bits), C.byref(bpc), C.byref(plen)
            )
            if st != 0 and attempt == 0 and want < cap and "payload buffer too small" in _ffi.last_error_string():
                want = min(cap, plen.value if plen.value > want else cap)  # the library reports the length it needs
                continue
            break
        check_status(st, "gpu_matrix_store_compact_bytes")
        self.is_ntt = False  # the store converts in place (MatrixSerde.cu:1108-1118)
        header = b"".join(
            [
                bytes([1, fmt]),
                _bincode_varint(self.level),
                _bincode_varint(self.nrow),
                _bincode_varint(self.ncol),
                _bincode_varint(max_bits.value),
                _bincode_varint(bpc.value),
                _bincode_varint(plen.value),
            ]
        )
        start = head_room - len(header)
        C.memmove(base + start, header, len(header))
        total = len(header) + plen.value
        return memoryview((C.c_ubyte * total).from_address(base + start)).cast("B")

    @classmethod
    def zero_compact_bytes(cls, params, nrow, ncol, level, is_ntt, max_coeff_bits) -> bytes:
        """Compact bytes of a zero matrix without touching the device (gpu_dcrt_poly.rs:1681-1710)."""
        assert level < params.crt_depth(), "invalid level for compact zero matrix"
        max_coeff_bits = max(int(max_coeff_bits), 1)
        bytes_per_coeff = -(-max_coeff_bits // 8)
        coeff_count = nrow * ncol * params.ring_dimension()
        payload_len = -(-coeff_count * max_coeff_bits // 8)
        fmt = GPU_POLY_FORMAT_EVAL if is_ntt else GPU_POLY_FORMAT_COEFF
        header = b"".join([bytes([1, fmt]), _bincode_varint(level), _bincode_varint(nrow), _bincode_varint(ncol),
                           _bincode_varint(max_coeff_bits), _bincode_varint(bytes_per_coeff), _bincode_varint(payload_len)])
        return header + bytes(payload_len)

    @classmethod
    def from_poly_vec(cls, params, rows) -> "GpuDCRTPolyMatrix":
        """`from_poly_vec` (gpu_dcrt_poly.rs:1092-1115): entries are transformed to EVAL and copied in."""
        if not rows:
            return cls.new_empty(params, 0, 0)
        nrow, ncol = len(rows), len(rows[0])
        if ncol == 0:
            return cls.new_empty(params, nrow, 0)
        first = rows[0][0].inner if hasattr(rows[0][0], "inner") else rows[0][0]
        out = cls(params, nrow, ncol, first.level, True)
        for i, row in enumerate(rows):
            assert len(row) == ncol, "row length mismatch in from_poly_vec"
            for j, poly in enumerate(row):
                m = poly.inner if hasattr(poly, "inner") else poly
                assert m.params == params, "params mismatch in from_poly_vec entry"
                assert m.level == first.level, "level mismatch in from_poly_vec entry"
                out.copy_block_from(m.ensure_eval(), i, j, 0, 0, 1, 1)
        return out

    def modulus_switch(self, new_modulus: int) -> "GpuDCRTPolyMatrix":
        """`modulus_switch` (gpu_dcrt_poly.rs:1352-1372): a host round trip, as in the reference - every coefficient c
        becomes floor(c * new_modulus / Q) mod new_modulus (src/element/finite_ring.rs:22-26); params are unchanged."""
        from .poly import GpuDCRTPoly

        Q = self.params.modulus()
        rows = []
        for row in self.coeffs():
            rows.append([GpuDCRTPoly.from_coeffs(self.params, [(c * new_modulus // Q) % new_modulus for c in poly]) for poly in row])
        return GpuDCRTPolyMatrix.from_poly_vec(self.params, rows)

    @classmethod
    def from_compact_bytes(cls, params, data: bytes) -> "GpuDCRTPolyMatrix":
        """gpu_dcrt_poly.rs:1004-1044."""
        version, fmt = data[0], data[1]
        assert version == 1, f"Unsupported compact matrix version: {version}"
        assert fmt in (GPU_POLY_FORMAT_COEFF, GPU_POLY_FORMAT_EVAL), f"Invalid compact matrix format tag: {fmt}"
        pos = 2
        level, pos = _bincode_read_varint(data, pos)
        nrow, pos = _bincode_read_varint(data, pos)
        ncol, pos = _bincode_read_varint(data, pos)
        max_bits, pos = _bincode_read_varint(data, pos)
        bpc, pos = _bincode_read_varint(data, pos)
        plen, pos = _bincode_read_varint(data, pos)
        payload = data[pos : pos + plen]
        assert len(payload) == plen and pos + plen == len(data), "truncated compact bytes"
        assert level < params.crt_depth(), f"invalid compact matrix level: {level}"
        assert bpc == (max_bits + 7) // 8, "compact bytes_per_coeff mismatch"
        out = cls(params, nrow, ncol, level, False)
        buf = C.cast(C.c_char_p(payload if plen else b"\0"), C.POINTER(C.c_uint8))  # no copy: the call is synchronous
        st = _ffi.lib().gpu_matrix_load_compact_bytes(out.raw, buf, plen, max_bits)
        check_status(st, "gpu_matrix_load_compact_bytes")
        out.is_ntt = False
        if fmt == GPU_POLY_FORMAT_EVAL:
            out.ntt_all_in_place()
        return out

    def to_cpu_staging_bytes(self) -> bytes:
        """RNS snapshot framing (gpu_dcrt_poly.rs:1046-1061): (1u8, nrow, ncol, level, is_ntt, bytes_per_poly, bytes)."""
        raw = memoryview(self.to_rns()).cast("B")  # joined below without an intermediate copy
        return b"".join(
            [
                bytes([1]),
                _bincode_varint(self.nrow),
                _bincode_varint(self.ncol),
                _bincode_varint(self.level),
                bytes([1 if self.is_ntt else 0]),
                _bincode_varint(self._bytes_per_poly()),
                _bincode_varint(len(raw)),
                raw,
            ]
        )

    into_cpu_staging_bytes = to_cpu_staging_bytes  # the consuming form (gpu_dcrt_poly.rs:1046)

    @classmethod
    def from_cpu_staging_bytes(cls, params, data: bytes) -> "GpuDCRTPolyMatrix":
        assert data[0] == 1, "Unsupported GPU matrix RNS staging version"
        pos = 1
        nrow, pos = _bincode_read_varint(data, pos)
        ncol, pos = _bincode_read_varint(data, pos)
        level, pos = _bincode_read_varint(data, pos)
        is_ntt = bool(data[pos])
        pos += 1
        bpp, pos = _bincode_read_varint(data, pos)
        blen, pos = _bincode_read_varint(data, pos)
        n = params.ring_dimension()
        assert bpp == (level + 1) * n * 8 and blen == nrow * ncol * bpp
        arr = np.frombuffer(data, dtype="<u8", count=blen // 8, offset=pos).reshape(nrow, ncol, level + 1, n)
        return cls.from_rns(params, arr, is_ntt)

    # ------------------------------------------------------------------ domain
    def ntt_all_in_place(self) -> None:
        if self.nrow == 0 or self.ncol == 0 or self.is_ntt:
            self.is_ntt = True
            return
        self._touch()
        check_status(_ffi.lib().gpu_matrix_ntt_all(self.raw), "gpu_matrix_ntt_all")
        self.is_ntt = True

    def intt_all_in_place(self) -> None:
        if self.nrow == 0 or self.ncol == 0 or not self.is_ntt:
            return
        self._touch()
        check_status(_ffi.lib().gpu_matrix_intt_all(self.raw), "gpu_matrix_intt_all")
        self.is_ntt = False

    def into_coeff_domain(self) -> "GpuDCRTPolyMatrix":
        self.intt_all_in_place()
        return self

    def ensure_coeff(self) -> "GpuDCRTPolyMatrix":
        if not self.is_ntt:
            return self
        return self.clone().into_coeff_domain()

    def ensure_eval(self) -> "GpuDCRTPolyMatrix":
        if self.is_ntt:
            return self
        out = self.clone()
        out.ntt_all_in_place()
        return out

    # ------------------------------------------------------------------ structure
    def clone(self) -> "GpuDCRTPolyMatrix":
        out = GpuDCRTPolyMatrix(self.params, self.nrow, self.ncol, self.level, self.is_ntt)
        if self.nrow and self.ncol:
            check_status(_ffi.lib().gpu_matrix_copy(out.raw, self.raw), "gpu_matrix_copy")
        return out

    def to_params(self, params) -> "GpuDCRTPolyMatrix":
        """Replica of this matrix in the context of `params` (another device's, usually): one peer copy over
        xGMI instead of the reference's `to_cpu_staging_bytes` -> `from_cpu_staging_bytes` host round trip
        (src/lookup/ggh15/pubkey_gpu.rs:153-196)."""
        if params.ctx_raw().value == self.params.ctx_raw().value:
            return self.clone()
        out = object.__new__(GpuDCRTPolyMatrix)
        raw = C.c_void_p()
        check_status(_ffi.lib().gpupoly_matrix_copy_to_context(params.ctx_raw(), self.raw, C.byref(raw)),
                     "gpupoly_matrix_copy_to_context")
        out.params, out.nrow, out.ncol, out.level, out.is_ntt, out.raw = params, self.nrow, self.ncol, self.level, self.is_ntt, raw
        out._parent, out._version = None, 0
        out._finalizer = weakref.finalize(out, _ffi.lib().gpu_matrix_destroy, raw)
        return out

    def size(self):
        return self.nrow, self.ncol

    def row_size(self) -> int:
        return self.nrow

    def col_size(self) -> int:
        return self.ncol

    def copy_block_from(self, src, dst_row, dst_col, src_row, src_col, rows, cols) -> None:
        if rows == 0 or cols == 0:
            return
        self._touch()
        st = _ffi.lib().gpu_matrix_copy_block(self.raw, src.raw, dst_row, dst_col, src_row, src_col, rows, cols)
        check_status(st, "gpu_matrix_copy_block")

    def add_block_from(self, src, dst_row, dst_col, src_row, src_col, rows, cols) -> None:
        assert self.params == src.params and self.level == src.level and self.is_ntt == src.is_ntt
        if rows == 0 or cols == 0:
            return
        self._touch()
        st = _ffi.lib().gpu_matrix_add_block(self.raw, src.raw, dst_row, dst_col, src_row, src_col, rows, cols)
        check_status(st, "gpu_matrix_add_block")
        self.is_ntt = src.is_ntt

    def add_rows_from(self, dst_row, lhs, rhs) -> None:
        """self[dst_row : dst_row + lhs.nrow] = lhs + rhs in one pass (gpupoly_matrix_add_rows)."""
        assert lhs.size() == rhs.size() and lhs.ncol == self.ncol and dst_row + lhs.nrow <= self.nrow
        if lhs.is_ntt != rhs.is_ntt:
            rhs = rhs.ensure_eval() if lhs.is_ntt else rhs.ensure_coeff()
        self._touch()
        check_status(_ffi.lib().gpupoly_matrix_add_rows(self.raw, dst_row, lhs.raw, rhs.raw), "gpupoly_matrix_add_rows")
        self.is_ntt = lhs.is_ntt

    def row_view(self, row_start, row_end) -> "GpuDCRTPolyMatrix":
        """Rows [row_start, row_end) as a matrix that shares this one's storage (gpupoly_matrix_row_view): an operand
        without the slice's copy.  The view keeps its parent alive; writes through either are seen by both."""
        assert 0 <= row_start <= row_end <= self.nrow
        raw = C.c_void_p()
        check_status(_ffi.lib().gpupoly_matrix_row_view(self.raw, row_start, row_end - row_start, C.byref(raw)), "gpupoly_matrix_row_view")
        v = object.__new__(GpuDCRTPolyMatrix)
        v.params, v.nrow, v.ncol, v.level, v.is_ntt, v.raw = self.params, row_end - row_start, self.ncol, self.level, self.is_ntt, raw
        v._parent = self
        v._version = 0
        v._finalizer = weakref.finalize(v, _ffi.lib().gpu_matrix_destroy, raw)
        return v

    def ntt_add_rows_from(self, dst_row, coeff, addend, consume: bool = False) -> None:
        """self[dst_row : dst_row + coeff.nrow] = NTT(coeff) + addend (gpupoly_matrix_ntt_add_rows): `coeff` holds
        coefficients, `addend` is EVAL; one pass where the fused kernel exists.  consume: the caller gives `coeff` up
        (do not use it afterwards) - where no fused kernel exists it is then transformed in place instead of copied."""
        assert coeff.size() == addend.size() and coeff.ncol == self.ncol and dst_row + coeff.nrow <= self.nrow
        assert not coeff.is_ntt, "ntt_add_rows_from takes a coefficient-domain matrix"
        addend = addend.ensure_eval()
        self._touch()
        st = _ffi.lib().gpupoly_matrix_ntt_add_rows(self.raw, dst_row, coeff.raw, addend.raw, 1 if consume else 0)
        check_status(st, "gpupoly_matrix_ntt_add_rows")
        self.is_ntt = True

    def slice(self, row_start, row_end, col_start, col_end) -> "GpuDCRTPolyMatrix":
        nrow, ncol = row_end - row_start, col_end - col_start
        out = GpuDCRTPolyMatrix(self.params, nrow, ncol, self.level, self.is_ntt)
        out.copy_block_from(self, 0, 0, row_start, col_start, nrow, ncol)
        return out

    def slice_rows(self, start, end):
        return self.slice(start, end, 0, self.ncol)

    def slice_columns(self, start, end):
        return self.slice(0, self.nrow, start, end)

    def entry(self, i, j):
        from .poly import GpuDCRTPoly

        return GpuDCRTPoly(self.slice(i, i + 1, j, j + 1))

    def set_entry(self, i, j, elem) -> None:
        src = elem.inner if hasattr(elem, "inner") else elem
        # convert domains first so the whole-matrix retag of copy_block stays harmless
        # (gpu_dcrt_poly.rs:1122-1132)
        src = src.ensure_eval() if self.is_ntt else src.ensure_coeff()
        self.copy_block_from(src, i, j, 0, 0, 1, 1)

    def get_row(self, i):
        return [self.entry(i, j) for j in range(self.ncol)]

    def get_column(self, j):
        return [self.entry(i, j) for i in range(self.nrow)]

    def transpose(self) -> "GpuDCRTPolyMatrix":
        """One launch (gpupoly_matrix_transpose); the reference loops nrow*ncol single-entry copy_block calls
        (gpu_dcrt_poly.rs:1190-1199)."""
        out = GpuDCRTPolyMatrix(self.params, self.ncol, self.nrow, self.level, self.is_ntt)
        if self.nrow and self.ncol:
            check_status(_ffi.lib().gpupoly_matrix_transpose(out.raw, self.raw), "gpupoly_matrix_transpose")
        return out

    def _same_domain(self, others):
        for o in others:
            assert o.params == self.params and o.level == self.level, "concat requires same params/level"
        return [o.ensure_eval() if self.is_ntt else o.ensure_coeff() for o in others]

    def concat_columns(self, others) -> "GpuDCRTPolyMatrix":
        others = self._same_domain(others)
        for o in others:
            assert o.nrow == self.nrow, "concat_columns requires same row count"
        ncol = self.ncol + sum(o.ncol for o in others)
        out = GpuDCRTPolyMatrix(self.params, self.nrow, ncol, self.level, self.is_ntt)
        off = 0
        for m in [self] + others:
            out.copy_block_from(m, 0, off, 0, 0, m.nrow, m.ncol)
            off += m.ncol
        out.is_ntt = self.is_ntt
        return out

    def concat_rows(self, others) -> "GpuDCRTPolyMatrix":
        others = self._same_domain(others)
        for o in others:
            assert o.ncol == self.ncol, "concat_rows requires same column count"
        nrow = self.nrow + sum(o.nrow for o in others)
        out = GpuDCRTPolyMatrix(self.params, nrow, self.ncol, self.level, self.is_ntt)
        off = 0
        for m in [self] + others:
            out.copy_block_from(m, off, 0, 0, 0, m.nrow, m.ncol)
            off += m.nrow
        out.is_ntt = self.is_ntt
        return out

    def concat_diag(self, others) -> "GpuDCRTPolyMatrix":
        others = self._same_domain(others)
        nrow = self.nrow + sum(o.nrow for o in others)
        ncol = self.ncol + sum(o.ncol for o in others)
        out = GpuDCRTPolyMatrix._new_zero_with_state(self.params, nrow, ncol, self.level, self.is_ntt)
        ro = co = 0
        for m in [self] + others:
            out.copy_block_from(m, ro, co, 0, 0, m.nrow, m.ncol)
            ro += m.nrow
            co += m.ncol
        out.is_ntt = self.is_ntt
        return out

    def tensor(self, other) -> "GpuDCRTPolyMatrix":
        """Kronecker product (gpu_dcrt_poly.rs:1225-1252: entry, mul_scalar, copy_block per entry there); one launch
        here.  As in the reference the blocks come out of mul_scalar, i.e. the result is EVAL."""
        assert self.params == other.params and self.level == other.level and self.is_ntt == other.is_ntt
        out = GpuDCRTPolyMatrix(self.params, self.nrow * other.nrow, self.ncol * other.ncol, self.level, self.is_ntt)
        if 0 in (self.nrow, self.ncol, other.nrow, other.ncol):
            return out
        lhs, rhs = self.ensure_eval(), other.ensure_eval()
        st = _ffi.lib().gpupoly_matrix_tensor(out.raw, lhs.raw, rhs.raw)
        check_status(st, "gpupoly_matrix_tensor")
        out.is_ntt = True
        return out

    def vectorize_columns(self) -> "GpuDCRTPolyMatrix":
        out = GpuDCRTPolyMatrix(self.params, self.nrow * self.ncol, 1, self.level, self.is_ntt)
        for j in range(self.ncol):
            out.copy_block_from(self, j * self.nrow, 0, 0, j, self.nrow, 1)
        return out

    # ------------------------------------------------------------------ arithmetic
    def _check_binop(self, rhs, what):
        assert self.params == rhs.params, f"{what} requires same params"
        assert self.level == rhs.level, f"{what} requires same level"
        assert self.is_ntt == rhs.is_ntt, f"{what} requires same domain"
        assert (self.nrow, self.ncol) == (rhs.nrow, rhs.ncol), f"{what} requires same dimensions"

    def add_in_place(self, rhs) -> None:
        self._check_binop(rhs, "add_in_place")
        if self.nrow == 0 or self.ncol == 0:
            return
        self._touch()
        check_status(_ffi.lib().gpu_matrix_add(self.raw, self.raw, rhs.raw), "gpu_matrix_add")
        self.is_ntt = rhs.is_ntt

    def sub_in_place(self, rhs) -> None:
        self._check_binop(rhs, "sub_in_place")
        if self.nrow == 0 or self.ncol == 0:
            return
        self._touch()
        check_status(_ffi.lib().gpu_matrix_sub(self.raw, self.raw, rhs.raw), "gpu_matrix_sub")
        self.is_ntt = rhs.is_ntt

    def _binop(self, rhs, fn, what):
        """out = self (+|-) rhs through the three-operand ABI call into a fresh matrix: three passes over memory.  The
        reference's `&a + &b` clones a and adds in place (gpu_dcrt_poly.rs:1731-1739): five passes."""
        self._check_binop(rhs, what)
        out = GpuDCRTPolyMatrix(self.params, self.nrow, self.ncol, self.level, self.is_ntt)
        if self.nrow == 0 or self.ncol == 0:
            return out
        check_status(fn(out.raw, self.raw, rhs.raw), what)
        out.is_ntt = rhs.is_ntt
        return out

    def __add__(self, rhs):
        return self._binop(rhs, _ffi.lib().gpu_matrix_add, "gpu_matrix_add")

    def __sub__(self, rhs):
        return self._binop(rhs, _ffi.lib().gpu_matrix_sub, "gpu_matrix_sub")

    def __neg__(self):
        """one pass (gpupoly_matrix_neg); the reference uploads zeros, clones and subtracts (gpu_dcrt_poly.rs:1890-1897)"""
        out = GpuDCRTPolyMatrix(self.params, self.nrow, self.ncol, self.level, self.is_ntt)
        if self.nrow == 0 or self.ncol == 0:
            return out
        check_status(_ffi.lib().gpupoly_matrix_neg(out.raw, self.raw), "gpupoly_matrix_neg")
        return out

    def mul_scalar(self, scalar) -> "GpuDCRTPolyMatrix":
        """`mul_scalar` (gpu_dcrt_poly.rs:1770-1790)."""
        s = scalar.inner if hasattr(scalar, "inner") else scalar
        lhs = self.ensure_eval()
        s = s.ensure_eval()
        out = GpuDCRTPolyMatrix(self.params, self.nrow, self.ncol, self.level, True)
        if self.nrow == 0 or self.ncol == 0:
            return out
        check_status(_ffi.lib().gpu_matrix_mul_scalar(out.raw, lhs.raw, s.raw), "gpu_matrix_mul_scalar")
        return out

    def mul_scalar_intt(self, scalar) -> "GpuDCRTPolyMatrix":
        """INTT(self o scalar) in one kernel (extension: the product rides in the inverse transform's load)."""
        s = scalar.inner if hasattr(scalar, "inner") else scalar
        lhs = self.ensure_eval()
        s = s.ensure_eval()
        out = GpuDCRTPolyMatrix(self.params, self.nrow, self.ncol, self.level, False)
        if self.nrow == 0 or self.ncol == 0:
            return out
        check_status(_ffi.lib().gpupoly_matrix_mul_scalar_intt(out.raw, lhs.raw, s.raw), "gpupoly_matrix_mul_scalar_intt")
        return out

    def __mul__(self, rhs):
        from .poly import GpuDCRTPoly

        if isinstance(rhs, GpuDCRTPoly):
            return self.mul_scalar(rhs)
        return self._mul_internal(rhs)

    __matmul__ = __mul__

    def _mul_internal(self, rhs) -> "GpuDCRTPolyMatrix":
        """`mul_internal` (gpu_dcrt_poly.rs:1792-1815)."""
        assert self.ncol == rhs.nrow, f"matrix multiply shape mismatch: ({self.nrow},{self.ncol}) x ({rhs.nrow},{rhs.ncol})"
        assert self.params == rhs.params, "mul requires same params"
        assert self.level == rhs.level, "mul requires same level"
        assert self.is_ntt and rhs.is_ntt, "mul requires NTT domain"
        out = GpuDCRTPolyMatrix(self.params, self.nrow, rhs.ncol, self.level, True)
        if self.nrow == 0 or rhs.ncol == 0:
            return out
        check_status(_ffi.lib().gpu_matrix_mul(out.raw, self.raw, rhs.raw), "gpu_matrix_mul")
        return out

    def __eq__(self, other):
        if not isinstance(other, GpuDCRTPolyMatrix):
            return NotImplemented
        if (
            self.params != other.params
            or (self.nrow, self.ncol) != (other.nrow, other.ncol)
            or self.level != other.level
            or self.is_ntt != other.is_ntt
        ):
            return False
        if self.raw.value == other.raw.value:
            return True
        eq = C.c_int(0)
        check_status(_ffi.lib().gpu_matrix_equal(self.raw, other.raw, C.byref(eq)), "gpu_matrix_equal")
        return eq.value != 0

    __hash__ = None

    # ------------------------------------------------------------------ decomposition
    def _decompose_from(self, src, out_nrow, small) -> "GpuDCRTPolyMatrix":
        out = GpuDCRTPolyMatrix.new_empty(self.params, out_nrow, self.ncol)
        fn = _ffi.lib().gpu_matrix_decompose_base_small if small else _ffi.lib().gpu_matrix_decompose_base
        check_status(fn(src.raw, self.params.base_bits(), out.raw), "gpu_matrix_decompose_base")
        return out

    def decompose(self) -> "GpuDCRTPolyMatrix":
        # an EVAL source goes to the library as it is: its coefficients are produced by an out-of-place inverse
        # transform into a scratch block (decompose.hip), not by clone() + in-place INTT (gpu_dcrt_poly.rs:1227-1233)
        return self._decompose_from(self, self.nrow * self.params.modulus_digits(), False)

    def decompose_owned(self) -> "GpuDCRTPolyMatrix":
        self.intt_all_in_place()
        return self._decompose_from(self, self.nrow * self.params.modulus_digits(), False)

    def small_decompose(self) -> "GpuDCRTPolyMatrix":
        k = -(-self.params.crt_bits() // self.params.base_bits())
        return self._decompose_from(self, self.nrow * k, True)

    def small_decompose_owned(self) -> "GpuDCRTPolyMatrix":
        self.intt_all_in_place()
        k = -(-self.params.crt_bits() // self.params.base_bits())
        return self._decompose_from(self, self.nrow * k, True)

    @classmethod
    def small_decomposed_identity_chunk(cls, params, size, chunk_idx, chunk_count, scalar_by_digit):
        """gpu_dcrt_poly.rs:1296-1326."""
        assert chunk_count > 0 and len(scalar_by_digit) == chunk_count
        assert chunk_idx < chunk_count
        polys = [p.inner.ensure_eval() if hasattr(p, "inner") else p.ensure_eval() for p in scalar_by_digit]
        row = polys[0].concat_columns(polys[1:]) if len(polys) > 1 else polys[0]
        out = cls.new_empty(params, size, size)
        st = _ffi.lib().gpu_matrix_fill_small_decomposed_identity_chunk(out.raw, row.raw, chunk_idx)
        check_status(st, "gpu_matrix_fill_small_decomposed_identity_chunk")
        return out

    @staticmethod
    def mul_batch(lhss, rhss) -> list:
        """[l * r for l, r in zip(lhss, rhss)] through `gpupoly_matrix_mul_batch`: the independent products of one
        circuit level in one call (small ones in one launch; src/circuit/poly_circuit/eval.rs:269 issues one per gate)."""
        assert len(lhss) == len(rhss)
        if not lhss:
            return []
        ls = [m.ensure_eval() for m in lhss]  # converted copies stay referenced until the call returns
        rs = [m.ensure_eval() for m in rhss]
        outs = []
        for l_, r_ in zip(ls, rs):
            assert l_.params == r_.params and l_.level == r_.level and l_.ncol == r_.nrow, "mul_batch: operand mismatch"
            outs.append(GpuDCRTPolyMatrix(l_.params, l_.nrow, r_.ncol, l_.level, True))
        arr = lambda ms: (C.c_void_p * len(ms))(*[m.raw.value if hasattr(m.raw, "value") else m.raw for m in ms])
        st = _ffi.lib().gpupoly_matrix_mul_batch(arr(outs), arr(ls), arr(rs), len(outs))
        check_status(st, "gpupoly_matrix_mul_batch")
        return outs

    @staticmethod
    def eval_gates(gates) -> list:
        """One level of independent circuit gates through `gpupoly_batch` (src/circuit/poly_circuit/eval.rs:269-345
        issues one ABI call per gate).  `gates` = [(kind, lhs, rhs_or_None), ...] with kind in {"mul", "add", "sub",
        "mul_scalar", "neg", "decompose", "mul_decompose"}; returns the gate outputs in order."""
        if not gates:
            return []
        kinds = {"mul": _ffi.GPUPOLY_OP_MUL, "add": _ffi.GPUPOLY_OP_ADD, "sub": _ffi.GPUPOLY_OP_SUB,
                 "mul_scalar": _ffi.GPUPOLY_OP_MUL_SCALAR, "neg": _ffi.GPUPOLY_OP_NEG,
                 "decompose": _ffi.GPUPOLY_OP_DECOMPOSE, "mul_decompose": _ffi.GPUPOLY_OP_MUL_DECOMPOSE}
        params = gates[0][1].params
        k = params.modulus_digits()
        ops = (_ffi.GpuBatchOp * len(gates))()
        outs, keep = [], []
        for i, (kind, lhs, rhs) in enumerate(gates):
            code = kinds[kind]
            if kind in ("mul", "mul_scalar", "mul_decompose"):
                lhs = lhs.ensure_eval()
                if kind != "mul_decompose":
                    rhs = rhs.ensure_eval()
            elif kind in ("add", "sub"):
                assert lhs.is_ntt == rhs.is_ntt, "add / sub gates need operands in one domain"
            if kind == "mul":
                out = GpuDCRTPolyMatrix(params, lhs.nrow, rhs.ncol, lhs.level, True)
            elif kind == "decompose":
                out = GpuDCRTPolyMatrix(params, lhs.nrow * k, lhs.ncol, lhs.level, True)
            elif kind == "mul_decompose":
                out = GpuDCRTPolyMatrix(params, lhs.nrow, rhs.ncol, lhs.level, True)
            else:
                out = GpuDCRTPolyMatrix(params, lhs.nrow, lhs.ncol, lhs.level, lhs.is_ntt)
            keep.append((lhs, rhs))
            ops[i].kind, ops[i].out, ops[i].lhs = code, out.raw, lhs.raw
            ops[i].rhs = rhs.raw if rhs is not None else None
            outs.append(out)
        check_status(_ffi.lib().gpupoly_batch(ops, len(gates), params.base_bits()), "gpupoly_batch")
        for (kind, lhs, rhs), out in zip(gates, outs):
            out.is_ntt = True if kind in ("mul", "mul_scalar", "decompose", "mul_decompose") else (rhs if rhs is not None else lhs).is_ntt
        return outs

    # ---- PolyMatrix trait defaults the GPU wrapper inherits (src/matrix/mod.rs:185-345) ------------------------
    def decompose_chunk(self, chunk_idx, chunk_count) -> "GpuDCRTPolyMatrix":
        assert chunk_count > 0, "decompose_chunk chunk_count must be > 0"
        assert chunk_idx < chunk_count, f"decompose_chunk chunk_idx out of range: chunk_idx={chunk_idx}, chunk_count={chunk_count}"
        full = self.decompose()
        assert full.nrow == self.nrow * chunk_count, f"decompose_chunk expected decomposed row count {self.nrow * chunk_count} but got {full.nrow}"
        return full.slice(chunk_idx * self.nrow, (chunk_idx + 1) * self.nrow, 0, self.ncol)

    def small_decompose_chunk(self, chunk_idx, chunk_count) -> "GpuDCRTPolyMatrix":
        assert chunk_count > 0, "small_decompose_chunk chunk_count must be > 0"
        assert chunk_idx < chunk_count, f"small_decompose_chunk chunk_idx out of range: chunk_idx={chunk_idx}, chunk_count={chunk_count}"
        full = self.small_decompose()
        assert full.nrow == self.nrow * chunk_count, f"small_decompose_chunk expected decomposed row count {self.nrow * chunk_count} but got {full.nrow}"
        return full.slice(chunk_idx * self.nrow, (chunk_idx + 1) * self.nrow, 0, self.ncol)

    @classmethod
    def small_decomposed_identity_chunk_from_scalar(cls, params, size, scalar, chunk_idx, chunk_count):
        """gpu_dcrt_poly.rs:1328-1350: the scalar's small digits, then one fill per chunk."""
        dec = cls.identity(params, 1, scalar).small_decompose()
        assert dec.size() == (chunk_count, 1), "scalar small decomposition shape mismatch in small_decomposed_identity_chunk_from_scalar"
        by_digit = [dec.entry(d, 0) for d in range(chunk_count)]
        return cls.small_decomposed_identity_chunk(params, size, chunk_idx, chunk_count, by_digit)

    @classmethod
    def unit_column_vector(cls, params, size, index) -> "GpuDCRTPolyMatrix":
        from .poly import GpuDCRTPoly

        assert index < size, "unit column index must be in range"
        col = [[GpuDCRTPoly.const_one(params) if i == index else GpuDCRTPoly.const_zero(params)] for i in range(size)]
        return cls.from_poly_vec(params, col)

    @classmethod
    def unit_row_vector(cls, params, size, index) -> "GpuDCRTPolyMatrix":
        from .poly import GpuDCRTPoly

        row = [GpuDCRTPoly.const_one(params) if j == index else GpuDCRTPoly.const_zero(params) for j in range(size)]
        return cls.from_poly_vec(params, [row])

    def block_entries(self, rows: range, cols: range) -> list:
        assert rows.start <= rows.stop <= self.nrow and cols.start <= cols.stop <= self.ncol, "block range out of bounds"
        return [[self.entry(i, j) for j in cols] for i in rows]

    def concat_rows_owned(self, others) -> "GpuDCRTPolyMatrix":
        return self.concat_rows(others)

    def concat_columns_owned(self, others) -> "GpuDCRTPolyMatrix":
        return self.concat_columns(others)

    def concat_diag_owned(self, others) -> "GpuDCRTPolyMatrix":
        return self.concat_diag(others)

    def mul_tensor_identity(self, other, identity_size) -> "GpuDCRTPolyMatrix":
        """self * (I (x) other) (gpu_dcrt_poly.rs:1374-1390): one extension call, products written in place."""
        assert self.ncol == other.nrow * identity_size
        if not mul_decompose_column_chunk_width_is_set():
            out = GpuDCRTPolyMatrix.new_empty(self.params, self.nrow, other.ncol * identity_size)
            if self.nrow == 0 or out.ncol == 0:
                return out
            lhs, rhs = self.ensure_eval(), other.ensure_eval()  # held: a converted copy must outlive the call
            st = _ffi.lib().gpupoly_matrix_mul_tensor_identity(out.raw, lhs.raw, rhs.raw, identity_size)
            check_status(st, "gpupoly_matrix_mul_tensor_identity")
            return out
        w = other.nrow
        slices = [self.slice(0, self.nrow, i * w, (i + 1) * w)._mul_internal(other) for i in range(identity_size)]
        return slices[0].concat_columns(slices[1:])

    def get_column_matrix_decompose(self, j) -> "GpuDCRTPolyMatrix":
        return self.slice(0, self.nrow, j, j + 1).decompose_owned()

    def mul_tensor_identity_decompose(self, other, identity_size) -> "GpuDCRTPolyMatrix":
        """self * (I (x) G^-1(other)) (gpu_dcrt_poly.rs:1392-1412).  The extension builds G^-1(other) once for all
        identity blocks; the reference's per-block, per-column loop runs when its chunk switch is set."""
        k = self.params.modulus_digits()
        assert self.ncol == other.nrow * identity_size * k
        if not mul_decompose_column_chunk_width_is_set():
            out = GpuDCRTPolyMatrix.new_empty(self.params, self.nrow, other.ncol * identity_size)
            if self.nrow == 0 or out.ncol == 0:
                return out
            lhs = self.ensure_eval()
            st = _ffi.lib().gpupoly_matrix_mul_tensor_identity_decompose(
                out.raw, lhs.raw, other.raw, identity_size, self.params.base_bits()
            )
            check_status(st, "gpupoly_matrix_mul_tensor_identity_decompose")
            return out
        w = other.nrow * k
        outs = []
        for i in range(identity_size):
            sl = self.slice(0, self.nrow, i * w, (i + 1) * w)
            for j in range(other.ncol):
                outs.append(sl._mul_internal(other.get_column_matrix_decompose(j)))
        return outs[0].concat_columns(outs[1:])

    def mul_decompose(self, other) -> "GpuDCRTPolyMatrix":
        """S * G^-1(B), column-chunked (gpu_dcrt_poly.rs:1414-1493)."""
        k = self.params.modulus_digits()
        assert self.ncol == other.nrow * k
        assert self.params == other.params
        ncol = other.ncol
        out = GpuDCRTPolyMatrix.new_empty(self.params, self.nrow, ncol)
        if self.nrow == 0 or ncol == 0:
            return out
        if not mul_decompose_column_chunk_width_is_set():
            # one ABI call: digits generated inside the forward transform, all columns at once, S read once
            # (gpupoly_matrix_mul_decompose).  The reference's column-chunk loop below (chunk
            # width 1 by default, re-reading S per chunk) runs only when its env switch is set explicitly.
            lhs = self.ensure_eval()
            st = _ffi.lib().gpupoly_matrix_mul_decompose(out.raw, lhs.raw, other.raw, self.params.base_bits())
            check_status(st, "gpupoly_matrix_mul_decompose")
            return out
        width = min(mul_decompose_column_chunk_width(), ncol)
        for c0 in range(0, ncol, width):
            c1 = min(c0 + width, ncol)
            dec = other.slice(0, other.nrow, c0, c1).decompose_owned()
            prod = self._mul_internal(dec)
            out.copy_block_from(prod, 0, c0, 0, 0, self.nrow, c1 - c0)
        return out

    def mul_decompose_small(self, other) -> "GpuDCRTPolyMatrix":
        k = -(-self.params.crt_bits() // self.params.base_bits())
        assert self.ncol == other.nrow * k
        ncol = other.ncol
        out = GpuDCRTPolyMatrix.new_empty(self.params, self.nrow, ncol)
        if self.nrow == 0 or ncol == 0:
            return out
        if not mul_decompose_column_chunk_width_is_set():
            lhs = self.ensure_eval()
            st = _ffi.lib().gpupoly_matrix_mul_decompose_small(out.raw, lhs.raw, other.raw, self.params.base_bits())
            check_status(st, "gpupoly_matrix_mul_decompose_small")
            return out
        width = min(mul_decompose_column_chunk_width(), ncol)
        for c0 in range(0, ncol, width):
            c1 = min(c0 + width, ncol)
            dec = other.slice(0, other.nrow, c0, c1).small_decompose_owned()
            prod = self._mul_internal(dec)
            out.copy_block_from(prod, 0, c0, 0, 0, self.nrow, c1 - c0)
        return out

    # ------------------------------------------------------------------ sampling entry points
    @classmethod
    def sample_distribution(cls, params, nrow, ncol, dist: int, sigma: float, seed: GpuRngSeed):
        out = cls.new_empty(params, nrow, ncol)
        if nrow == 0 or ncol == 0:
            return out
        check_status(_ffi.lib().gpu_matrix_sample_distribution(out.raw, dist, sigma, seed), "gpu_matrix_sample_distribution")
        return out

    @classmethod
    def sample_distribution_columns(cls, params, nrow, total_ncol, col_start, col_len, dist, sigma, seed):
        assert col_start + col_len <= total_ncol, "sample_distribution_columns range out of bounds"
        out = cls.new_empty(params, nrow, col_len)
        if nrow == 0 or col_len == 0:
            return out
        st = _ffi.lib().gpu_matrix_sample_distribution_columns(out.raw, dist, sigma, seed, total_ncol, col_start)
        check_status(st, "gpu_matrix_sample_distribution_columns")
        return out

    @classmethod
    def sample_distribution_decomposed(cls, params, nrow, ncol, dist: int, sigma: float, seed: GpuRngSeed, small=False):
        """G^-1 (or the small G^-1) of `sample_distribution(params, nrow, ncol, ...)` through the
        `gpupoly_matrix_sample_decomposed` extension: the samples never leave the coefficient domain."""
        k = -(-params.crt_bits() // params.base_bits()) if small else params.modulus_digits()
        out = cls.new_empty(params, nrow * k, ncol)
        if nrow == 0 or ncol == 0:
            return out
        st = _ffi.lib().gpupoly_matrix_sample_decomposed(out.raw, dist, sigma, seed, params.base_bits(), 1 if small else 0)
        check_status(st, "gpupoly_matrix_sample_decomposed")
        return out

    def gauss_samp_gq_arb_base(self, c: float, dgg_stddev: float, seed: GpuRngSeed, coeff_out: bool = False) -> "GpuDCRTPolyMatrix":
        """Consumes self (gpu_dcrt_poly.rs:509-528).  coeff_out: leave the digits as coefficients (the entry point
        finishes in whatever format the output matrix is tagged with) for a caller that transforms them itself
        (`ntt_add_rows_from`)."""
        out = GpuDCRTPolyMatrix(self.params, self.nrow * self.params.modulus_digits(), self.ncol, self.params.crt_depth() - 1,
                                not coeff_out)
        self.intt_all_in_place()
        self._touch()
        st = _ffi.lib().gpu_matrix_gauss_samp_gq_arb_base(self.raw, self.params.base_bits(), c, dgg_stddev, seed, out.raw)
        check_status(st, "gpu_matrix_gauss_samp_gq_arb_base")
        return out

    @staticmethod
    def create_p1_covariance_cache(a_mat, b_mat, d_mat, sigma, s, dgg_stddev) -> GpuP1CovarianceCache:
        raw = C.c_void_p()
        st = _ffi.lib().gpu_matrix_create_p1_covariance_cache(a_mat.raw, b_mat.raw, d_mat.raw, sigma, s, dgg_stddev, C.byref(raw))
        check_status(st, "gpu_matrix_create_p1_covariance_cache")
        return GpuP1CovarianceCache(raw, a_mat.params)

    @staticmethod
    def sample_p1_full_cached(cache: GpuP1CovarianceCache, tp2, seed: GpuRngSeed) -> "GpuDCRTPolyMatrix":
        out = GpuDCRTPolyMatrix.new_empty(tp2.params, tp2.nrow, tp2.ncol)
        if tp2.nrow == 0 or tp2.ncol == 0:
            return out
        tp2.intt_all_in_place()
        check_status(_ffi.lib().gpu_matrix_sample_p1_full_cached(cache.raw, tp2.raw, seed, out.raw), "gpu_matrix_sample_p1_full_cached")
        return out

    # ---- several independently seeded requests in one launch (gpupoly_*_segments; include/gpupoly.h) ------------------
    @staticmethod
    def _segment_args(seeds, seg_cols):
        assert len(seeds) == len(seg_cols) and seeds, "one seed per segment"
        return (GpuRngSeed * len(seeds))(*seeds), (C.c_size_t * len(seg_cols))(*seg_cols), len(seeds)

    @classmethod
    def sample_distribution_segments(cls, params, nrow, seg_cols, dist: int, sigma: float, seeds) -> "GpuDCRTPolyMatrix":
        """[S_0 | S_1 | ...] with S_j == sample_distribution(params, nrow, seg_cols[j], dist, sigma, seeds[j]), one launch.
        Raises GpuPolyError (text contains "unsupported") where the library has no segmented form."""
        out = cls.new_empty(params, nrow, sum(seg_cols))
        arr, cols, n = cls._segment_args(seeds, seg_cols)
        check_status(_ffi.lib().gpupoly_matrix_sample_distribution_segments(out.raw, dist, sigma, arr, cols, n),
                     "gpupoly_matrix_sample_distribution_segments")
        return out

    @staticmethod
    def sample_p1_full_cached_segments(cache: GpuP1CovarianceCache, tp2, seeds, seg_cols) -> "GpuDCRTPolyMatrix":
        """`sample_p1_full_cached` over column segments with a seed each (tp2 is taken to the coefficient domain in place)."""
        out = GpuDCRTPolyMatrix.new_empty(tp2.params, tp2.nrow, tp2.ncol)
        arr, cols, n = GpuDCRTPolyMatrix._segment_args(seeds, seg_cols)
        tp2.intt_all_in_place()
        check_status(_ffi.lib().gpupoly_matrix_sample_p1_full_cached_segments(cache.raw, tp2.raw, arr, cols, n, out.raw),
                     "gpupoly_matrix_sample_p1_full_cached_segments")
        return out

    def gauss_samp_gq_arb_base_segments(self, c: float, dgg_stddev: float, seeds, seg_cols, coeff_out: bool = False) -> "GpuDCRTPolyMatrix":
        """`gauss_samp_gq_arb_base` over column segments with a seed each; consumes self."""
        out = GpuDCRTPolyMatrix(self.params, self.nrow * self.params.modulus_digits(), self.ncol, self.params.crt_depth() - 1,
                                not coeff_out)
        arr, cols, n = self._segment_args(seeds, seg_cols)
        self.intt_all_in_place()
        self._touch()
        st = _ffi.lib().gpupoly_matrix_gauss_samp_gq_arb_base_segments(self.raw, self.params.base_bits(), c, dgg_stddev, arr, cols, n, out.raw)
        check_status(st, "gpupoly_matrix_gauss_samp_gq_arb_base_segments")
        return out

    @staticmethod
    def _raw_array(ms):
        return (C.c_void_p * len(ms))(*[m.raw.value if hasattr(m.raw, "value") else m.raw for m in ms])

    @classmethod
    def concat_columns_of(cls, blocks) -> "GpuDCRTPolyMatrix":
        """[blocks[0] | blocks[1] | ...] in one launch per 64 blocks (`gpupoly_matrix_concat_columns`); the blocks share one
        domain (converted copies are made like `concat_columns` does)."""
        blocks = [blocks[0]] + blocks[0]._same_domain(blocks[1:])
        first = blocks[0]
        for b in blocks:
            assert b.nrow == first.nrow and b.level == first.level and b.params == first.params, "concat_columns_of: block mismatch"
        out = cls(first.params, first.nrow, sum(b.ncol for b in blocks), first.level, first.is_ntt)
        check_status(_ffi.lib().gpupoly_matrix_concat_columns(out.raw, cls._raw_array(blocks), len(blocks)), "gpupoly_matrix_concat_columns")
        return out

    def split_columns(self, widths) -> list:
        """the column blocks of the given widths, in order, in one launch per 64 blocks (`gpupoly_matrix_split_columns`)"""
        assert sum(widths) == self.ncol, "split_columns: widths must add up to the column count"
        outs = [GpuDCRTPolyMatrix(self.params, self.nrow, w, self.level, self.is_ntt) for w in widths]
        check_status(_ffi.lib().gpupoly_matrix_split_columns(self.raw, self._raw_array(outs), len(outs)), "gpupoly_matrix_split_columns")
        return outs

    def __repr__(self):
        return f"GpuDCRTPolyMatrix({self.nrow}x{self.ncol}, level={self.level}, is_ntt={self.is_ntt})"
