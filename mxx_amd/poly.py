"""`GpuDCRTPoly` — a polynomial is a 1x1 `GpuDCRTPolyMatrix` (src/poly/dcrt/gpu.rs:707-1100).

`Poly` trait: src/poly/mod.rs:79-198.
"""
from __future__ import annotations

import numpy as np

from .matrix import GpuDCRTPolyMatrix
from .params import GpuDCRTPolyParams


class GpuDCRTPoly:
    __slots__ = ("inner",)

    def __init__(self, inner: GpuDCRTPolyMatrix):
        inner_rows, inner_cols = inner.size()
        assert inner_rows == 1 and inner_cols == 1, "matrix must be 1x1 for poly operation"
        self.inner = inner

    # ---- constructors --------------------------------------------------------------
    @classmethod
    def _from_residues(cls, params: GpuDCRTPolyParams, residues: np.ndarray, eval_format: bool) -> "GpuDCRTPoly":
        L, n = residues.shape
        return cls(GpuDCRTPolyMatrix.from_rns(params, residues.reshape(1, 1, L, n), eval_format))

    @classmethod
    def from_biguints(cls, params, coeffs) -> "GpuDCRTPoly":
        """Coefficients (python ints) -> COEFF residues -> NTT (gpu.rs:930-933,841-857)."""
        n = params.ring_dimension()
        assert len(coeffs) <= n
        moduli = params.moduli()
        res = np.zeros((len(moduli), n), dtype=np.uint64)
        for l, q in enumerate(moduli):
            res[l, : len(coeffs)] = [int(c) % q for c in coeffs]
        p = cls._from_residues(params, res, False)
        p.inner.ntt_all_in_place()
        return p

    from_coeffs = from_biguints

    @classmethod
    def from_biguints_eval(cls, params, slots) -> "GpuDCRTPoly":
        # Rust-side oddity kept as is: slot values are loaded as coefficients and
        # transformed (gpu.rs:935-939)
        return cls.from_biguints(params, slots)

    @classmethod
    def from_u32s(cls, params, coeffs) -> "GpuDCRTPoly":
        return cls.from_biguints(params, [int(c) for c in coeffs])

    @classmethod
    def from_bool_vec(cls, params, coeffs) -> "GpuDCRTPoly":
        return cls.from_biguints(params, [1 if c else 0 for c in coeffs])

    @classmethod
    def from_biguint_to_constant(cls, params, value: int) -> "GpuDCRTPoly":
        return cls.from_biguints(params, [value])

    from_usize_to_constant = from_biguint_to_constant
    from_elem_to_constant = from_biguint_to_constant  # FinRingElem -> its value (gpu.rs:1021-1023)

    @classmethod
    def const_zero(cls, params):
        return cls.from_biguints(params, [0])

    @classmethod
    def const_one(cls, params):
        return cls.from_biguints(params, [1])

    @classmethod
    def const_minus_one(cls, params):
        return cls.from_biguints(params, [params.modulus() - 1])

    @classmethod
    def const_max(cls, params):
        return cls.from_biguints(params, [params.modulus() - 1] * params.ring_dimension())

    @classmethod
    def from_power_of_base_to_constant(cls, params, k: int):
        return cls.from_biguints(params, [1 << (params.base_bits() * k)])

    @classmethod
    def from_usize_to_lsb(cls, params, value: int):
        n = params.ring_dimension()
        return cls.from_biguints(params, [(value >> i) & 1 for i in range(n)])

    @classmethod
    def from_u64_vecs(cls, params, coeffs) -> "GpuDCRTPoly":
        """coeffs[i] = the residues of coefficient i, limb by limb (gpu.rs:758-788): the limb count of the longest entry
        sets the level, missing residues are 0, the polynomial stays in COEFF form."""
        n = params.ring_dimension()
        assert len(coeffs) <= n, f"coeffs length must be <= ring dimension (got {len(coeffs)}, expected <= {n})"
        num_limbs = max([len(v) for v in coeffs] + [1])
        assert num_limbs <= params.crt_depth(), "coeff limb count exceeds CRT depth"
        flat = np.zeros((num_limbs, n), dtype=np.uint64)
        for i, c in enumerate(coeffs):
            flat[: len(c), i] = c
        return cls._from_residues(params, flat, False)

    @classmethod
    def from_inner(cls, inner: GpuDCRTPolyMatrix) -> "GpuDCRTPoly":
        return cls(inner)

    # ---- accessors -----------------------------------------------------------------
    def params(self) -> GpuDCRTPolyParams:
        return self.inner.params

    params_ref = params

    def level(self) -> int:
        return self.inner.level

    def ntt_in_place(self) -> None:
        self.inner.ntt_all_in_place()

    def store_rns_bytes(self, bytes_out, fmt: int) -> None:
        """gpu.rs:790-795: the whole buffer is one polynomial's stride."""
        if len(bytes_out) == 0:
            return
        self.inner.store_rns_bytes(bytes_out, len(bytes_out), fmt)

    def assert_compatible(self, other: "GpuDCRTPoly") -> None:
        assert self.level() == other.level(), "GPU polynomials must have the same level"
        assert self.params() == other.params(), "GPU params must match"

    def is_ntt(self) -> bool:
        return self.inner.is_ntt

    def coeffs(self) -> list[int]:
        return self.inner.coeffs()[0][0]

    def ensure_coeff_domain(self) -> "GpuDCRTPoly":
        return GpuDCRTPoly(self.inner.ensure_coeff())

    def ensure_eval_domain(self) -> "GpuDCRTPoly":
        return GpuDCRTPoly(self.inner.ensure_eval())

    def clone(self) -> "GpuDCRTPoly":
        return GpuDCRTPoly(self.inner.clone())

    def decompose_base(self) -> list["GpuDCRTPoly"]:
        dec = self.inner.decompose()
        return [dec.entry(i, 0) for i in range(dec.nrow)]

    @classmethod
    def from_decomposed(cls, params, decomposed) -> "GpuDCRTPoly":
        """sum_i 2^i * decomposed[i] (gpu.rs:941-949)."""
        acc = cls.const_zero(params)
        for i, bit_poly in enumerate(decomposed):
            acc = acc + bit_poly * cls.from_biguint_to_constant(params, 1 << i)
        return acc

    @classmethod
    def from_compact_bytes(cls, params, data: bytes) -> "GpuDCRTPoly":
        """gpu.rs:951-957: the bytes of a 1x1 matrix."""
        mat = GpuDCRTPolyMatrix.from_compact_bytes(params, data)
        assert mat.size() == (1, 1), "GpuDCRTPoly compact bytes must decode to 1x1 matrix"
        return mat.entry(0, 0)

    def to_compact_bytes(self) -> bytes:
        return self.inner.to_compact_bytes()

    def const_coeff_u64(self) -> int:
        """Constant coefficient through `gpu_matrix_store_const_coeff_batch` + CRT (gpu.rs:1103-1120)."""
        poly = self.inner.ensure_coeff()
        residues = [int(v) for v in poly.store_const_coeff_words().reshape(-1)]
        moduli = poly.params.moduli()[: poly.level + 1]
        Q = 1
        for q in moduli:
            Q *= q
        acc = 0
        for r, q in zip(residues, moduli):
            Qi = Q // q
            acc += Qi * pow(Qi, -1, q) * r
        value = acc % Q
        if value >> 64:
            raise OverflowError(f"constant coefficient does not fit in u64: {value}")
        return value

    def extract_bits_with_threshold(self) -> list[bool]:
        """coefficient in [q/4, 3q/4) -> True (gpu.rs:1070-1081; quarter = (q/2) >> 1)."""
        quarter = (self.inner.params.modulus() // 2) >> 1
        return [quarter <= c < 3 * quarter for c in self.coeffs()]

    def to_bool_vec(self) -> list[bool]:
        out = []
        for c in self.coeffs():
            if c not in (0, 1):
                raise ValueError(f"Coefficient is not 0 or 1: {c}")
            out.append(c == 1)
        return out

    # ---- arithmetic -------------------------------------------------------------------
    def _pair(self, other):
        assert self.inner.params == other.inner.params
        # both operands go to EVAL, as in the reference (gpu.rs:1123-1143)
        return self.inner.ensure_eval(), other.inner.ensure_eval()

    def __add__(self, other):
        a, b = self._pair(other)
        return GpuDCRTPoly(a + b)

    def __sub__(self, other):
        a, b = self._pair(other)
        return GpuDCRTPoly(a - b)

    def __neg__(self):
        return GpuDCRTPoly(-self.inner)

    def __mul__(self, other):
        if isinstance(other, GpuDCRTPolyMatrix):
            return other.mul_scalar(self)
        return GpuDCRTPoly(self.inner.ensure_eval().mul_scalar(other))

    def __eq__(self, other):
        if not isinstance(other, GpuDCRTPoly):
            return NotImplemented
        if self.inner.params != other.inner.params or self.inner.level != other.inner.level:
            return False
        # compare across domains by going to COEFF first (gpu.rs:864-879)
        return self.inner.ensure_coeff() == other.inner.ensure_coeff()

    __hash__ = None
