"""ctypes front-end of the CPU oracle (oracle/oracle.c, oracle/oracle_sampling.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke(); never by the product package `mxx_amd`.

All matrices use the reference's wire layout `[poly][limb][n]` of little-endian
u64 residues with poly = row*cols + col (src/poly/dcrt/gpu.rs:758-788,
src/matrix/gpu_dcrt_poly.rs:781-816 of the reference), as numpy uint64 arrays of
shape (rows, cols, L, n).
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_SAN_PATH = os.path.join(_HERE, "liboracle_asan.so")
_CFLAGS = ["-fopenmp", "-fPIC", "-std=gnu11", "-ffp-contract=off"]  # oracle/Makefile builds with the same flags


def build_sanitized() -> str:
    """AddressSanitizer + UndefinedBehaviorSanitizer build of the same sources (the reference runs
    its native code under sanitizers on the CPU).  tests/test_oracle_sanitized.py re-runs the oracle's own CPU tests on
    it in a child process (MXX_ORACLE_LIB selects the library, libasan is preloaded)."""
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_sampling.c")]
    deps = srcs + [os.path.join(_HERE, "..", "mxx_amd", "csrc", "detmath.h")]
    if not os.path.exists(_SAN_PATH) or any(os.path.getmtime(s) > os.path.getmtime(_SAN_PATH) for s in deps):
        subprocess.check_call(
            ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all"]
            + _CFLAGS + ["-shared", "-o", _SAN_PATH] + srcs + ["-lm"]
        )
    return _SAN_PATH


def build(force: bool = False) -> str:
    if os.environ.get("MXX_ORACLE_LIB"):  # a prebuilt variant (the sanitizer build) chosen by the caller
        return os.environ["MXX_ORACLE_LIB"]
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_sampling.c")]
    deps = srcs + [os.path.join(_HERE, "..", "mxx_amd", "csrc", "detmath.h")]  # Box-Muller's log / cos, shared text
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in deps)
    ):
        subprocess.check_call(
            ["gcc", "-O3"] + _CFLAGS + ["-shared", "-o", _LIB_PATH]
            + srcs
            + ["-lm"]
        )
    return _LIB_PATH


def use_native_build() -> str:
    """bench.py's cpu_baseline leg: compile the same sources with -march=native for THIS host into a
    temp dir and make lib() load that build (the in-tree liboracle.so is a portable build that also
    travels to other machines).  Must be called before the first lib()."""
    global _LIB_PATH, _lib
    import tempfile

    if "liboracle_native" in _LIB_PATH:
        return _LIB_PATH
    os.environ.pop("MXX_ORACLE_LIB", None)  # the timed baseline is never the sanitizer build
    if _lib is not None:
        raise RuntimeError("use_native_build() must run before the library is first loaded")
    out = os.path.join(tempfile.mkdtemp(prefix="oracle_native_"), "liboracle_native.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle.c", "oracle_sampling.c")]
    subprocess.check_call(
        ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-shared", "-o", out]
        + srcs
        + ["-lm"]
    )
    _LIB_PATH = out
    return out


_lib = None
_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orc_mulmod.restype = C.c_uint64
        L.orc_mulmod.argtypes = [C.c_uint64] * 3
        L.orc_powmod.restype = C.c_uint64
        L.orc_powmod.argtypes = [C.c_uint64] * 3
        L.orc_invmod.restype = C.c_uint64
        L.orc_invmod.argtypes = [C.c_uint64] * 2
        L.orc_is_prime.restype = C.c_int
        L.orc_is_prime.argtypes = [C.c_uint64]
        L.orc_gen_crt_basis.restype = C.c_int
        L.orc_gen_crt_basis.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, _u64p]
        L.orc_min_primitive_root.restype = C.c_uint64
        L.orc_min_primitive_root.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_ntt_tables.restype = None
        L.orc_ntt_tables.argtypes = [C.c_uint64, C.c_uint32, _u64p, _u64p, _u64p]
        L.orc_ntt.restype = None
        L.orc_ntt.argtypes = [_u64p, C.c_uint32, C.c_uint64, C.c_int, C.c_int]
        L.orc_matrix_ntt.restype = None
        L.orc_matrix_ntt.argtypes = [_u64p, C.c_size_t, C.c_uint32, C.c_uint32, _u64p, C.c_int]
        L.orc_pointwise.restype = None
        L.orc_pointwise.argtypes = [C.c_int, _u64p, _u64p, _u64p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u64p]
        for name in ("orc_matmul", "orc_matmul_fast"):
            f = getattr(L, name)
            f.restype = None
            f.argtypes = [_u64p, _u64p, _u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u64p]
        L.orc_negacyclic_schoolbook.restype = None
        L.orc_negacyclic_schoolbook.argtypes = [_u64p, _u64p, _u64p, C.c_uint32, C.c_uint64]
        L.orc_decompose.restype = None
        L.orc_decompose.argtypes = [_u64p, _u64p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u64p, C.c_uint32, C.c_int]
        L.orc_fill_gadget.restype = None
        L.orc_fill_gadget.argtypes = [_u64p, C.c_size_t, C.c_uint32, C.c_uint32, _u64p, C.c_uint32, C.c_int]
        L.orc_ring_mul_batch.restype = None
        L.orc_ring_mul_batch.argtypes = [_u64p, _u64p, _u64p, C.c_size_t, C.c_uint32, C.c_uint32, _u64p]
        L.orc_set_threads.restype = None
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_crt_bits.restype = C.c_uint32
        L.orc_crt_bits.argtypes = [_u64p, C.c_uint32]
        _bind_sampling(L)
    return _lib


def _bind_sampling(L):
    if not hasattr(L, "orc_chacha20_block"):
        return
    u32p = C.POINTER(C.c_uint32)
    L.orc_chacha20_block.restype = None
    L.orc_chacha20_block.argtypes = [u32p, u32p]
    L.orc_rng_stream.restype = None
    L.orc_rng_stream.argtypes = [_u64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, _u64p, C.c_size_t]
    L.orc_sample_distribution.restype = None
    L.orc_sample_distribution.argtypes = [
        _u64p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u64p,
        C.c_int, C.c_double, _u64p,
    ]
    L.orc_sample_distribution_refkey.restype = None
    L.orc_sample_distribution_refkey.argtypes = L.orc_sample_distribution.argtypes
    L.orc_karney.restype = None
    L.orc_karney.argtypes = [_u64p, C.c_uint64, C.c_double, C.c_double, _i64p, C.c_size_t]
    L.orc_karney_ties.restype = C.c_ulonglong
    L.orc_karney_ties.argtypes = []
    L.orc_gauss_samp_gq.restype = None
    L.orc_gauss_samp_gq.argtypes = [
        _u64p, _u64p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u64p, C.c_uint32, C.c_double, _u64p,
    ]
    L.orc_p1_covariance.restype = None
    L.orc_p1_covariance.argtypes = [
        _u64p, _u64p, _u64p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint64,
        C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double),
    ]
    L.orc_sample_p1.restype = None
    L.orc_sample_p1.argtypes = [
        _u64p, _u64p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, _u64p,
        C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, _u64p,
    ]


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def _mod(moduli) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(moduli, dtype=np.uint64))


# ----------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------
def gen_crt_basis(n: int, depth: int, bits: int) -> list[int]:
    out = np.zeros(depth, dtype=np.uint64)
    rc = lib().orc_gen_crt_basis(n, depth, bits, _p(out))
    if rc != 0:
        raise ValueError(f"orc_gen_crt_basis({n},{depth},{bits}) failed: {rc}")
    return [int(x) for x in out]


def min_primitive_root(q: int, order: int) -> int:
    return int(lib().orc_min_primitive_root(q, order))


def ntt_tables(q: int, n: int):
    fwd = np.zeros(n, dtype=np.uint64)
    inv = np.zeros(n, dtype=np.uint64)
    ninv = C.c_uint64(0)
    lib().orc_ntt_tables(q, n, _p(fwd), _p(inv), C.byref(ninv))
    return fwd, inv, int(ninv.value)


# ----------------------------------------------------------------------------
# transforms / arithmetic on (rows, cols, L, n) uint64 arrays
# ----------------------------------------------------------------------------
def ntt_vec(x: np.ndarray, q: int, inverse: bool = False, plain: bool = False) -> np.ndarray:
    y = np.ascontiguousarray(x, dtype=np.uint64).copy()
    lib().orc_ntt(_p(y), y.shape[-1], q, int(inverse), int(plain))
    return y


def matrix_ntt(m: np.ndarray, moduli, inverse: bool = False) -> np.ndarray:
    y = np.ascontiguousarray(m, dtype=np.uint64).copy()
    L, n = y.shape[-2], y.shape[-1]
    polys = y.size // (L * n)
    lib().orc_matrix_ntt(_p(y), polys, L, n, _p(_mod(moduli)), int(inverse))
    return y


def pointwise(op: str, a: np.ndarray, b: np.ndarray, moduli) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    L, n = a.shape[-2], a.shape[-1]
    polys = a.size // (L * n)
    bpolys = b.size // (L * n)
    assert bpolys in (1, polys)
    out = np.empty_like(a)
    lib().orc_pointwise({"add": 0, "sub": 1, "mul": 2}[op], _p(out), _p(a), _p(b), polys, bpolys, L, n, _p(_mod(moduli)))
    return out


def matmul(a: np.ndarray, b: np.ndarray, moduli, fast: bool = False) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    rows, inner, L, n = a.shape
    inner2, cols, L2, n2 = b.shape
    assert inner == inner2 and L == L2 and n == n2
    out = np.zeros((rows, cols, L, n), dtype=np.uint64)
    f = lib().orc_matmul_fast if fast else lib().orc_matmul
    f(_p(out), _p(a), _p(b), rows, inner, cols, L, n, _p(_mod(moduli)))
    return out


def negacyclic_schoolbook(a: np.ndarray, b: np.ndarray, q: int) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    lib().orc_negacyclic_schoolbook(_p(out), _p(a), _p(b), a.shape[-1], q)
    return out


def crt_bits(moduli) -> int:
    m = _mod(moduli)
    return int(lib().orc_crt_bits(_p(m), len(m)))


def digits_per_tower(moduli, base_bits: int) -> int:
    return -(-crt_bits(moduli) // base_bits)


def decompose(src_coeff: np.ndarray, moduli, base_bits: int, small: bool = False) -> np.ndarray:
    """COEFF in -> COEFF out, (rows*k, cols, L, n)."""
    src = np.ascontiguousarray(src_coeff, dtype=np.uint64)
    rows, cols, L, n = src.shape
    dpt = digits_per_tower(moduli, base_bits)
    k = dpt if small else dpt * L
    out = np.zeros((rows * k, cols, L, n), dtype=np.uint64)
    lib().orc_decompose(_p(out), _p(src), rows, cols, L, n, _p(_mod(moduli)), base_bits, int(small))
    return out


def gadget_matrix(size: int, moduli, n: int, base_bits: int, small: bool = False, eval_format: bool = True) -> np.ndarray:
    L = len(moduli)
    dpt = digits_per_tower(moduli, base_bits)
    k = dpt if small else dpt * L
    out = np.zeros((size, size * k, L, n), dtype=np.uint64)
    lib().orc_fill_gadget(_p(out), size, L, n, _p(_mod(moduli)), base_bits, int(small))
    return matrix_ntt(out, moduli) if eval_format else out


def ring_mul_batch(a: np.ndarray, b: np.ndarray, moduli) -> np.ndarray:
    """c = a*b in R_q for a batch of COEFF polys (consumes copies)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    b = np.ascontiguousarray(b, dtype=np.uint64).copy()
    L, n = a.shape[-2], a.shape[-1]
    polys = a.size // (L * n)
    c = np.empty_like(a)
    lib().orc_ring_mul_batch(_p(c), _p(a), _p(b), polys, L, n, _p(_mod(moduli)))
    return c


# ----------------------------------------------------------------------------
# CRT reconstruction (python ints; small cases only)
# ----------------------------------------------------------------------------
def crt_reconstruct(residues, moduli) -> int:
    """residues[l] mod moduli[l] -> integer in [0, Q)  (src/poly/mod.rs:45-76 of the reference)."""
    Q = 1
    for q in moduli:
        Q *= int(q)
    x = 0
    for r, q in zip(residues, moduli):
        q = int(q)
        Qi = Q // q
        x += int(r) * Qi * pow(Qi, -1, q)
    return x % Q


def splitmix64(seed: int, count: int) -> np.ndarray:
    """Synthetic-input generator named by SURVEY.md §8(d)."""
    out = np.empty(count, dtype=np.uint64)
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        out[:] = z ^ (z >> np.uint64(31))
    return out


def random_matrix(seed: int, rows: int, cols: int, moduli, n: int) -> np.ndarray:
    """i.i.d. uniform residues in [0,q_l), layout (rows, cols, L, n)."""
    L = len(moduli)
    raw = splitmix64(seed, rows * cols * L * n).reshape(rows, cols, L, n)
    q = np.asarray(moduli, dtype=np.uint64).reshape(1, 1, L, 1)
    return raw % q


# ----------------------------------------------------------------------------
# seeded samplers (oracle_sampling.c)
# ----------------------------------------------------------------------------
def _seed_words(seed) -> np.ndarray:
    if isinstance(seed, (bytes, bytearray)):
        assert len(seed) == 32
        return np.frombuffer(bytes(seed), dtype="<u8").astype(np.uint64).copy()
    if hasattr(seed, "words"):
        return np.array([int(w) for w in seed.words], dtype=np.uint64)
    return np.ascontiguousarray(seed, dtype=np.uint64)


def chacha20_block(state16) -> np.ndarray:
    s = np.ascontiguousarray(state16, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    u32p = C.POINTER(C.c_uint32)
    lib().orc_chacha20_block(s.ctypes.data_as(u32p), out.ctypes.data_as(u32p))
    return out


def rng_stream(seed, s0, s1, s2, tag, count) -> np.ndarray:
    out = np.zeros(count, dtype=np.uint64)
    lib().orc_rng_stream(_p(_seed_words(seed)), s0, s1, s2, tag, _p(out), count)
    return out


def karney_samples(seed, s0, mean, stddev, count) -> np.ndarray:
    out = np.zeros(count, dtype=np.int64)
    lib().orc_karney(_p(_seed_words(seed)), s0, float(mean), float(stddev), out.ctypes.data_as(_i64p), count)
    return out


def karney_ties() -> int:
    """Comparisons of Karney's sampler that tied on the 16-bit draw and needed the low bits, since the library was loaded."""
    return int(lib().orc_karney_ties())


DIST = {"uniform": 0, "gauss": 1, "bit": 2, "ternary": 3}


def sample_distribution(rows, cols, moduli, n, dist: str, sigma, seed, full_ncol=None, col_offset=0) -> np.ndarray:
    """COEFF residues (rows, cols, L, n) of the seeded sampler (before the NTT the ABI applies)."""
    L = len(moduli)
    out = np.zeros((rows, cols, L, n), dtype=np.uint64)
    full = cols if full_ncol is None else full_ncol
    lib().orc_sample_distribution(_p(out), rows, cols, full, col_offset, L, n, _p(_mod(moduli)), DIST[dist], float(sigma), _p(_seed_words(seed)))
    return out


def sample_distribution_refkey(rows, cols, moduli, n, dist: str, sigma, seed, full_ncol=None, col_offset=0) -> np.ndarray:
    """The same call with the reference device RNG's OWN keying (cuda/src/ChaCha.cu:104-167, MatrixSampling.cu:239-289):
    what libgpupoly must produce under MXX_HIP_RNG_COMPAT=reference."""
    L = len(moduli)
    out = np.zeros((rows, cols, L, n), dtype=np.uint64)
    full = cols if full_ncol is None else full_ncol
    lib().orc_sample_distribution_refkey(_p(out), rows, cols, full, col_offset, L, n, _p(_mod(moduli)), DIST[dist], float(sigma),
                                         _p(_seed_words(seed)))
    return out


def gauss_samp_gq(src_coeff: np.ndarray, moduli, base_bits: int, c: float, seed) -> np.ndarray:
    src = np.ascontiguousarray(src_coeff, dtype=np.uint64)
    rows, cols, L, n = src.shape
    k = digits_per_tower(moduli, base_bits) * L
    out = np.zeros((rows * k, cols, L, n), dtype=np.uint64)
    lib().orc_gauss_samp_gq(_p(out), _p(src), rows, cols, L, n, _p(_mod(moduli)), base_bits, float(c), _p(_seed_words(seed)))
    return out


def p1_covariance(a, b, d, moduli, sigma, s, dgg_stddev):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    d = np.ascontiguousarray(d, dtype=np.uint64)
    dd, _, L, n = a.shape
    m = 2 * dd
    sv = np.zeros((n, m), dtype=np.float64)
    up = np.zeros((n, m, m), dtype=np.float64)
    dp = C.POINTER(C.c_double)
    lib().orc_p1_covariance(_p(a), _p(b), _p(d), dd, L, n, int(moduli[0]), float(sigma), float(s), float(dgg_stddev), sv.ctypes.data_as(dp), up.ctypes.data_as(dp))
    return sv, up


def sample_p1(tp2_coeff, moduli, sqrt_var, update, c_scale, seed) -> np.ndarray:
    tp2 = np.ascontiguousarray(tp2_coeff, dtype=np.uint64)
    m, cols, L, n = tp2.shape
    out = np.zeros_like(tp2)
    dp = C.POINTER(C.c_double)
    sv = np.ascontiguousarray(sqrt_var, dtype=np.float64)
    up = np.ascontiguousarray(update, dtype=np.float64)
    lib().orc_sample_p1(_p(out), _p(tp2), m, cols, L, n, _p(_mod(moduli)), sv.ctypes.data_as(dp), up.ctypes.data_as(dp), float(c_scale), _p(_seed_words(seed)))
    return out


def centered(res: np.ndarray, q: int) -> np.ndarray:
    r = res.astype(np.int64)
    return np.where(r > q // 2, r - q, r)


def compact_payload(coeff_rns: np.ndarray, moduli):
    """Compact wire payload of a COEFF matrix (python ints; small cases only).

    Follows cuda/src/matrix/MatrixSerde.cu:280-456,1535-1627 of the reference: per
    coefficient (poly-major) the centred representative of the CRT value, |x| in the low
    w-1 bits and the sign in bit w-1, little-endian bit stream; w = 1 + max bit-length of
    |x| (0 for all-zero).  Returns (payload bytes, max_coeff_bits, bytes_per_coeff)."""
    rows, cols, L, n = coeff_rns.shape
    moduli = [int(q) for q in moduli[:L]]
    Q = 1
    for q in moduli:
        Q *= q
    weights = []
    for q in moduli:
        Qi = Q // q
        weights.append(Qi * pow(Qi, -1, q))
    vals = []
    for r in range(rows):
        for c in range(cols):
            for i in range(n):
                x = sum(int(coeff_rns[r, c, l, i]) * weights[l] for l in range(L)) % Q
                vals.append((Q - x, 1) if x > Q // 2 else (x, 0))
    width = max((m.bit_length() for m, _ in vals), default=0)
    width = width + 1 if width else 0
    stream = 0
    for idx, (m, s) in enumerate(vals):
        stream |= (m | (s << (width - 1))) << (idx * width) if width else 0
    nbytes = (len(vals) * width + 7) // 8
    return stream.to_bytes(nbytes, "little"), width, (width + 7) // 8


# ---- trapdoor generation and preimage, end to end on the CPU -------------------------------------
# Restates the reference's GPU preimage sequence (src/sampler/trapdoor/gpu.rs:69-80,202-215,228-369,
# 423-474; constants :15-27) with the functions above.  Matrices are EVAL residues (rows, cols, L, n).
def _seed_from(seed, salt: int):
    w = _seed_words(seed).copy()
    w[0] ^= np.uint64(0x9e3779b97f4a7c15 * (salt + 1) & 0xFFFFFFFFFFFFFFFF)
    return w


def preimage_params(moduli, n: int, base_bits: int, sigma: float, d: int):
    k = digits_per_tower(moduli, base_bits) * len(moduli)
    b = float(1 << base_bits)
    c = (b + 1.0) * sigma
    s = 1.8 * (b + 1.0) * sigma * sigma * (math.sqrt(d * n * k) + math.sqrt(2 * n) + 4.7)
    return k, c, s


def trapdoor_gen(moduli, n: int, base_bits: int, sigma: float, d: int, seed):
    """R, E ~ D_sigma^{d x dk}; A = [Abar | I_d | G - (Abar R + E)] (gpu.rs:202-215)."""
    k, _, _ = preimage_params(moduli, n, base_bits, sigma, d)
    L = len(moduli)
    r = matrix_ntt(sample_distribution(d, d * k, moduli, n, "gauss", sigma, _seed_from(seed, 0)), moduli)
    e = matrix_ntt(sample_distribution(d, d * k, moduli, n, "gauss", sigma, _seed_from(seed, 1)), moduli)
    abar = matrix_ntt(sample_distribution(d, d, moduli, n, "uniform", 0.0, _seed_from(seed, 2)), moduli)
    ident = np.zeros((d, d, L, n), dtype=np.uint64)
    for i in range(d):
        ident[i, i] = 1  # EVAL form of the constant 1
    g = gadget_matrix(d, moduli, n, base_bits)
    a1 = pointwise("sub", g, pointwise("add", matmul(abar, r, moduli, fast=True), e, moduli), moduli)
    return r, e, np.concatenate([abar, ident, a1], axis=1)


def preimage(moduli, n: int, base_bits: int, sigma: float, r, e, a, target, seed, cov=None):
    """x with a * x == target (gpu.rs:228-369).  `cov` = p1_covariance(...) of this trapdoor, if cached."""
    d, dk = r.shape[0], r.shape[1]
    cols = target.shape[1]
    k, c, s = preimage_params(moduli, n, base_bits, sigma, d)
    assert dk == d * k and a.shape[1] == dk + 2 * d
    re = np.concatenate([r, e], axis=0)
    if cov is None:
        rt, et = np.swapaxes(r, 0, 1), np.swapaxes(e, 0, 1)
        inv = lambda m: matrix_ntt(m, moduli, inverse=True)
        cov = p1_covariance(inv(matmul(r, rt, moduli, fast=True)), inv(matmul(r, et, moduli, fast=True)),
                            inv(matmul(e, et, moduli, fast=True)), moduli, c, s, sigma)
    sv, up = cov
    # the perturbation is sampled for ceil(cols / d) * d columns and cut back (gpu.rs:436-438, 288-292):
    # the padding changes the global polynomial indices the streams are keyed by
    padded = -(-cols // d) * d
    p2 = matrix_ntt(sample_distribution(dk, padded, moduli, n, "gauss", math.sqrt(s * s - c * c), _seed_from(seed, 3)), moduli)
    tp2 = matrix_ntt(matmul(re, p2, moduli, fast=True), moduli, inverse=True)
    p1 = matrix_ntt(sample_p1(tp2, moduli, sv, up, -(c * c) / (s * s - c * c), _seed_from(seed, 4)), moduli)
    p1, p2 = p1[:, :cols], p2[:, :cols]
    p_hat = pointwise("add", matmul(a[:, : 2 * d], p1, moduli, fast=True), matmul(a[:, 2 * d :], p2, moduli, fast=True), moduli)
    pert = matrix_ntt(pointwise("sub", target, p_hat, moduli), moduli, inverse=True)
    z = matrix_ntt(gauss_samp_gq(pert, moduli, base_bits, c, _seed_from(seed, 5)), moduli)
    top = pointwise("add", p1, matmul(re, z, moduli, fast=True), moduli)
    return np.concatenate([top, pointwise("add", p2, z, moduli)], axis=0)
