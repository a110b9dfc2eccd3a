"""Trapdoor generation and preimage sampling on the GPU.

Mirrors `GpuDCRTTrapdoor` / `GpuDCRTPolyTrapdoorSampler`
(src/sampler/trapdoor/gpu.rs:15-474; trait `PolyTrapdoorSampler`,
src/sampler/mod.rs:122-207) call for call: every step is one C-ABI entry of
libgpupoly on the context's stream, no host round trip until the caller reads.
"""
from __future__ import annotations

import math
import threading
import weakref

from .matrix import GpuDCRTPolyMatrix
from .sampler import DistType, GpuDCRTPolyUniformSampler, random_gpu_rng_seed, sample_gpu_matrix_with_seed

SPECTRAL_CONSTANT = 1.8  # trapdoor/gpu.rs:15


def preimage_c(base: int, sigma: float) -> float:
    return (base + 1.0) * sigma


def preimage_smoothing_parameter(base: int, sigma: float, d: int, n: int, k: int) -> float:
    return SPECTRAL_CONSTANT * (base + 1.0) * sigma * sigma * (math.sqrt(d * n * k) + math.sqrt(2 * n) + 4.7)


def p1_covariance_parameters(params, d: int, dgg_stddev: float):
    """(c, s, dgg_stddev) of the p1 covariance cache (trapdoor/gpu.rs:132-143)."""
    base = 1 << params.base_bits()
    n, k = params.ring_dimension(), params.modulus_digits()
    return preimage_c(base, dgg_stddev), preimage_smoothing_parameter(base, dgg_stddev, d, n, k), dgg_stddev


def compute_preimage_norm(ring_dim_sqrt: float, m_g: int, base: float, b_nrow=None, sigma=None) -> float:
    """The preimage coefficient bound the reference's tests check against (src/simulator/eval_error/evaluators.rs:679-700);
    restated in float64 (the reference uses BigDecimal; the quantities are a few thousand at most)."""
    sigma = 4.578 if sigma is None else sigma
    b_nrow = 1 if b_nrow is None else b_nrow
    term = math.sqrt(b_nrow) * ring_dim_sqrt * math.sqrt(m_g) + math.sqrt(2.0) * ring_dim_sqrt + 4.7
    return 1.8 * 6.5 * sigma * ((base + 1.0) * sigma) * term


def _coeff_cached(m: GpuDCRTPolyMatrix) -> GpuDCRTPolyMatrix:
    return m.clone().into_coeff_domain()


coeff_cached_matrix = _coeff_cached  # gpu.rs:28-30


class GpuPerturbationSamples:
    """(p1, p2) of `sample_pert_square_mat_gpu_native_parts` (gpu.rs:32-35)."""

    __slots__ = ("p1", "p2")

    def __init__(self, p1, p2):
        self.p1, self.p2 = p1, p2

    def __iter__(self):
        return iter((self.p1, self.p2))


class GpuDCRTTrapdoor:
    """R, E ~ D_sigma^{d x dk} plus the coefficient-domain RR^T, RE^T, EE^T caches (gpu.rs:46-80)."""

    def __init__(self, r: GpuDCRTPolyMatrix, e: GpuDCRTPolyMatrix):
        self.r = r
        self.e = e
        self.re = r.concat_rows([e])  # [R; E], the left factor of both tp2 = [R;E] p2 and [R;E] z
        rt, et = r.transpose(), e.transpose()
        self.a_mat_coeff = _coeff_cached(r * rt)
        self.b_mat_coeff = _coeff_cached(r * et)
        self.d_mat_coeff = _coeff_cached(e * et)
        self._p1_cache = None
        self._p1_lock = threading.Lock()
        self._stacked = None  # ([R; E; right], the public-matrix block it was built from): see stacked_left_factor
        self._replicas = {}   # context handle -> (trapdoor, public matrix, ...) in that context: see replica_for

    @classmethod
    def new(cls, params, size: int, sigma: float) -> "GpuDCRTTrapdoor":
        u = GpuDCRTPolyUniformSampler()
        k = params.modulus_digits()
        dist = DistType.GaussDist(sigma)
        return cls(u.sample_uniform(params, size, size * k, dist), u.sample_uniform(params, size, size * k, dist))

    def p1_covariance_cache(self, c: float, s: float, dgg_stddev: float):
        with self._p1_lock:
            if self._p1_cache is not None and self._p1_cache[0] == (c, s, dgg_stddev):
                return self._p1_cache[1]
            cache = GpuDCRTPolyMatrix.create_p1_covariance_cache(
                self.a_mat_coeff, self.b_mat_coeff, self.d_mat_coeff, c, s, dgg_stddev
            )
            self._p1_cache = ((c, s, dgg_stddev), cache)
            return cache

    get_or_create_p1_covariance_cache = p1_covariance_cache  # gpu.rs:137-200

    def public_matrix_parts(self, public_matrix: GpuDCRTPolyMatrix):
        """(left, right, [R; E; right]) of A = [left | right] for this trapdoor: the two column blocks the preimage multiplies
        by p1 / p2 and the stacked left factor of the one product over p2 that yields both [R;E] p2 and right p2.  Built once
        per public-matrix OBJECT and content version (callers keep a trapdoor with its matrix:
        `src/sampler/trapdoor/gpu.rs:228-369` slices A on every call) and reused while that object is alive and has not been
        rewritten in place through the mirror (`GpuDCRTPolyMatrix.content_version`: copy_block_from, add_block_from,
        load_rns*, the in-place transforms ... all bump it).  The cache pins one copy of A plus the stacked factor in device
        memory for the trapdoor's lifetime; `clear_public_matrix_cache()` releases it."""
        with self._p1_lock:
            if self._stacked is not None and self._stacked[0]() is public_matrix and self._stacked[2] == public_matrix.content_version():
                return self._stacked[1]
            d, p1_rows, p2_rows = public_matrix.row_size(), self.re.row_size(), self.re.col_size()
            left = public_matrix.slice(0, d, 0, p1_rows)
            right = public_matrix.slice(0, d, p1_rows, p1_rows + p2_rows)
            parts = (left, right, self.re.concat_rows([right]))
            self._stacked = (weakref.ref(public_matrix), parts, public_matrix.content_version())
            return parts

    def clear_public_matrix_cache(self) -> None:
        """drop the column blocks / stacked factor kept by `public_matrix_parts` (one A plus [R; E; right] of device memory)
        and the replicas kept by `replica_for`"""
        with self._p1_lock:
            self._stacked = None
            self._replicas = {}

    def replica_for(self, params, public_matrix: GpuDCRTPolyMatrix):
        """(trapdoor, public matrix) in the context of `params` - another stream on the same device (a worker context of
        `preimage_batched_sharded`) or another device: device-to-device copies of R, E and A plus the small products,
        made once per (context, public-matrix object and content version) and kept while this trapdoor lives."""
        key = params.ctx_raw().value
        with self._p1_lock:
            hit = self._replicas.get(key)
            if hit is not None and hit[2]() is public_matrix and hit[3] == public_matrix.content_version():
                return hit[0], hit[1]
        td, a = self.to_params(params), public_matrix.to_params(params)
        with self._p1_lock:
            self._replicas[key] = (td, a, weakref.ref(public_matrix), public_matrix.content_version())
        return td, a

    def to_params(self, params) -> "GpuDCRTTrapdoor":
        """Replica of the trapdoor on another device context: two peer copies (R, E) and the small products
        recomputed there - instead of trapdoor_to_bytes / trapdoor_from_bytes through the host
        (src/lookup/ggh15/pubkey_gpu.rs:153-196)."""
        return GpuDCRTTrapdoor(self.r.to_params(params), self.e.to_params(params))

    def to_compact_bytes(self) -> bytes:
        """R then E, each as u64-LE length + compact matrix bytes (gpu.rs:82-97)."""
        out = b""
        for m in (self.r, self.e):
            b = m.to_compact_bytes()
            out += len(b).to_bytes(8, "little") + b
        return out

    @classmethod
    def from_compact_bytes(cls, params, data: bytes):
        """gpu.rs:99-129; returns None on malformed input like the reference's Option."""
        mats, off = [], 0
        for _ in range(2):
            if off + 8 > len(data):
                return None
            ln = int.from_bytes(data[off : off + 8], "little")
            off += 8
            if off + ln > len(data):
                return None
            mats.append(GpuDCRTPolyMatrix.from_compact_bytes(params, data[off : off + ln]))
            off += ln
        if off != len(data):
            return None
        return cls(mats[0], mats[1])

    def __eq__(self, other):
        return isinstance(other, GpuDCRTTrapdoor) and self.r == other.r and self.e == other.e

    __hash__ = None


WORKER_DNUM = 0x5700  # dnum of worker contexts: a context's identity is (ring, moduli, base, devices, dnum)
_worker_params_cache: dict = {}
_worker_params_lock = threading.Lock()


def preimage_workers() -> int:
    """`MXX_PREIMAGE_WORKERS` (default 4; 1 = off): worker contexts per device for requests that do not share a trapdoor"""
    import os

    try:
        return max(1, min(8, int(os.environ.get("MXX_PREIMAGE_WORKERS", "4"))))
    except ValueError:
        return 4


def worker_params(params, k: int):
    """the k-th worker context beside `params` (k >= 1): same ring, moduli and device, its own stream and allocator"""
    from .params import GpuDCRTPolyParams

    key = (params.ctx_raw().value, k)
    with _worker_params_lock:
        for dead in [kk for kk, (owner, _) in _worker_params_cache.items() if owner() is None]:
            del _worker_params_cache[dead]  # the context a worker belonged to is gone: let the worker context go too
        hit = _worker_params_cache.get(key)
        if hit is not None and hit[0]() is params.ctx():
            return hit[1]
        pw = GpuDCRTPolyParams(params.ring_dimension(), params.moduli(), params.base_bits(), gpu_ids=params.gpu_ids(), dnum=WORKER_DNUM + k)
        _worker_params_cache[key] = (weakref.ref(params.ctx()), pw)
        return pw


TRAFFIC_BOUND_BYTES = 32 << 20  # tests set it to 0 to run the large-operand assembly at small sizes


def _traffic_bound(params, polys: int) -> bool:
    """A matrix of `polys` polynomials is large enough (32 MiB) for the passes over it, not the launches, to be the cost:
    the preimage then trades a few small launches for whole passes over its largest operands."""
    word = 4 if max(params.moduli()) < (1 << 31) else 8
    return polys * params.crt_depth() * params.ring_dimension() * word >= TRAFFIC_BOUND_BYTES


class GpuDCRTPolyTrapdoorSampler:
    def __init__(self, params, sigma: float):
        self.sigma = float(sigma)
        self.base = 1 << params.base_bits()
        self.c = preimage_c(self.base, self.sigma)

    @staticmethod
    def trapdoor_to_bytes(trapdoor: "GpuDCRTTrapdoor") -> bytes:
        return trapdoor.to_compact_bytes()

    @staticmethod
    def trapdoor_from_bytes(params, data: bytes):
        return GpuDCRTTrapdoor.from_compact_bytes(params, data)

    def trapdoor(self, params, size: int):
        """A = [A_bar | I | G - (A_bar R + E)] (gpu.rs:202-215)."""
        u = GpuDCRTPolyUniformSampler()
        td = GpuDCRTTrapdoor.new(params, size, self.sigma)
        a_bar = u.sample_uniform(params, size, size, DistType.FinRingDist())
        g = GpuDCRTPolyMatrix.gadget_matrix(params, size)
        a0 = a_bar.concat_columns([GpuDCRTPolyMatrix.identity(params, size)])
        a1 = g - (a_bar * td.r + td.e)
        return td, a0.concat_columns([a1])

    def _sample_pert(self, params, td: GpuDCRTTrapdoor, s, c, dgg_stddev, sigma_large, total_ncol):
        """`sample_pert_square_mat_gpu_native_parts` (gpu.rs:423-474): (p1, p2), the reference's signature."""
        return self._sample_pert_parts(params, td, s, c, dgg_stddev, sigma_large, total_ncol)[:2]

    def sample_pert_square_mat_gpu_native_parts(self, params, td, s, c, dgg_stddev, sigma_large, total_ncol):
        return GpuPerturbationSamples(*self._sample_pert(params, td, s, c, dgg_stddev, sigma_large, total_ncol))

    def sample_pert_square_mat_gpu_native(self, params, td, s, c, dgg_stddev, sigma_large, total_ncol):
        """p_hat = [p1; p2] cut to the `total_ncol` target columns (gpu.rs:502-541: the padding columns of the last
        d-block are sampled and skipped at assembly)."""
        p1, p2 = self._sample_pert(params, td, s, c, dgg_stddev, sigma_large, total_ncol)
        assert p1.col_size() >= total_ncol and p2.col_size() >= total_ncol, "p1 / p2 must include the target columns"
        p_hat = GpuDCRTPolyMatrix(params, p1.row_size() + p2.row_size(), total_ncol, p1.level, p1.is_ntt)
        p_hat.copy_block_from(p1, 0, 0, 0, 0, p1.row_size(), total_ncol)
        p_hat.copy_block_from(p2, p1.row_size(), 0, 0, 0, p2.row_size(), total_ncol)  # both samplers finish in EVAL
        return p_hat

    @staticmethod
    def _draw_seeds():
        """the three seeds of one preimage call, in the order the reference draws them: p2, p1, z"""
        return random_gpu_rng_seed(), random_gpu_rng_seed(), random_gpu_rng_seed()

    def _sample_pert_parts(self, params, td: GpuDCRTTrapdoor, s, c, dgg_stddev, sigma_large, total_ncol, stacked=None, seeds=None):
        """The same with the by-products the large-operand assembly reuses.  With `stacked` = [R; E; right] (right = the
        public matrix's columns over p2; `GpuDCRTTrapdoor.public_matrix_parts`) the product right * p2 the caller needs next
        rides in the same pass over p2 as [R;E] p2.  Returns p1, p2, [R;E] p2 (EVAL, kept for the final assembly) and
        right * p2 (or None); the last two are row views of the one product - no copies - and only the p1 sampler, which
        takes its argument to the coefficient domain in place, gets a slice of its own."""
        d, dk = td.r.row_size(), td.r.col_size()
        padded = -(-total_ncol // d) * d
        seed_p2 = seeds[0] if seeds is not None else random_gpu_rng_seed()
        p2 = sample_gpu_matrix_with_seed(params, dk, padded, DistType.GaussDist(sigma_large), seed_p2)
        rp2 = tp2_eval = None
        if stacked is not None and _traffic_bound(params, dk * padded):
            t = stacked * p2
            re_rows = td.re.row_size()
            tp2 = t.slice(0, re_rows, 0, padded)
            tp2_eval = t.row_view(0, re_rows)
            rp2 = t.row_view(re_rows, t.row_size())
        else:
            tp2 = td.re * p2
            if _traffic_bound(params, dk * padded):
                tp2_eval = tp2.clone()
        cache = td.p1_covariance_cache(c, s, dgg_stddev)
        p1 = GpuDCRTPolyMatrix.sample_p1_full_cached(cache, tp2, seeds[1] if seeds is not None else random_gpu_rng_seed())
        return p1, p2, tp2_eval, rp2

    def preimage(self, params, td: GpuDCRTTrapdoor, public_matrix, target, _seeds=None) -> GpuDCRTPolyMatrix:
        """x with public_matrix * x == target (gpu.rs:228-369).  `_seeds` = (p2, p1, z) seeds already drawn by a caller
        that batches requests (`preimage_many`); otherwise they are drawn here, in that order."""
        d = public_matrix.row_size()
        target_cols = target.col_size()
        assert target.row_size() == d, "Target matrix should have the same number of rows as the public matrix"
        n, k = params.ring_dimension(), params.modulus_digits()
        s = preimage_smoothing_parameter(self.base, self.sigma, d, n, k)
        dgg_large_std = math.sqrt(s * s - self.c * self.c)
        p1_rows, p2_rows = td.re.row_size(), td.re.col_size()
        assert public_matrix.col_size() == p1_rows + p2_rows, "public matrix columns must match perturbation rows"
        left, right, stacked = td.public_matrix_parts(public_matrix)
        p1, p2, tp2, rp2 = self._sample_pert_parts(params, td, s, self.c, self.sigma, dgg_large_std, target_cols, stacked, _seeds)
        seed_z = _seeds[2] if _seeds is not None else None
        assert (p1.row_size(), p2.row_size()) == (p1_rows, p2_rows)
        p_hat_image = (left * p1) + (rp2 if rp2 is not None else right * p2)
        if p_hat_image.col_size() != target_cols:
            p_hat_image = p_hat_image.slice_columns(0, target_cols)
        perturbed = target - p_hat_image
        out = GpuDCRTPolyMatrix(params, p1_rows + p2_rows, target_cols, p1.level, p1.is_ntt)
        if p1.col_size() == target_cols and tp2 is not None:
            # x = [p1 + [R;E] z ; p2 + z] (gpu.rs:340-362 transforms z, forms R z and E z, adds twice).  Here the
            # G-sampler's digits stay coefficients and ONE pass writes the bottom block NTT(z) + p2 into its place;
            # the top block follows from it, [R;E] z = [R;E] (p2 + z) - [R;E] p2, with the product read straight
            # from the output's rows and [R;E] p2 kept from the perturbation step - same residues, and z's
            # evaluation form (the largest matrix of the call) is never written or re-read.
            z = perturbed.gauss_samp_gq_arb_base(self.c, self.sigma, seed_z if seed_z is not None else random_gpu_rng_seed(), coeff_out=True)
            out.ntt_add_rows_from(p1_rows, z, p2, consume=True)
            re_x = td.re * out.row_view(p1_rows, p1_rows + p2_rows)
            out.add_rows_from(0, p1 - tp2, re_x)
            return out
        z_hat = perturbed.gauss_samp_gq_arb_base(self.c, self.sigma, seed_z if seed_z is not None else random_gpu_rng_seed())
        re_z = td.re * z_hat
        if p1.col_size() == target_cols:
            # small operands (launch-bound): the sums written straight into out's row blocks, 3 passes instead of 5
            out.add_rows_from(0, p1, re_z)
            out.add_rows_from(p1_rows, p2, z_hat)
            return out
        out.copy_block_from(p1, 0, 0, 0, 0, p1_rows, target_cols)
        out.copy_block_from(p2, p1_rows, 0, 0, 0, p2_rows, target_cols)
        out.add_block_from(re_z, 0, 0, 0, 0, re_z.row_size(), target_cols)
        out.add_block_from(z_hat, 2 * d, 0, 0, 0, z_hat.row_size(), target_cols)
        return out

    def preimage_reference_sequence(self, params, td: GpuDCRTTrapdoor, public_matrix, target) -> GpuDCRTPolyMatrix:
        """The SAME preimage through nothing but the reference's own call sequence (gpu.rs:228-369 and
        `sample_pert_square_mat_gpu_native_parts`, gpu.rs:423-474), i.e. what an unpatched mxx gets from this library: A is
        sliced and [R; E] concatenated on every call, tp2 is a product of its own, z comes back in EVAL form, R z and E z
        are two products, and the output is assembled with two copy_block + three add_block calls.  Only entry points the
        Rust side binds are used (44 `gpu_*` symbols; no `gpupoly_*` extension).  Same distribution and same seeds ->
        same residues as `preimage` (tests/test_gpu_preimage_batch.py); bench.py times it next to the extension sequence."""
        d = public_matrix.row_size()
        target_cols = target.col_size()
        assert target.row_size() == d, "Target matrix should have the same number of rows as the public matrix"
        n, k = params.ring_dimension(), params.modulus_digits()
        s = preimage_smoothing_parameter(self.base, self.sigma, d, n, k)
        dgg_large_std = math.sqrt(s * s - self.c * self.c)
        # sample_pert_square_mat_gpu_native_parts
        u = GpuDCRTPolyUniformSampler()
        dd, dk = td.r.row_size(), td.r.col_size()
        padded = -(-target_cols // dd) * dd
        p2 = u.sample_uniform(params, dk, padded, DistType.GaussDist(dgg_large_std))
        tp2 = td.r.concat_rows([td.e]) * p2
        cache = td.p1_covariance_cache(self.c, s, self.sigma)
        p1 = GpuDCRTPolyMatrix.sample_p1_full_cached(cache, tp2, random_gpu_rng_seed())
        # preimage
        p1_rows, p2_rows = p1.row_size(), p2.row_size()
        assert public_matrix.col_size() == p1_rows + p2_rows, "public matrix columns must match perturbation rows"
        public_left = public_matrix.slice(0, d, 0, p1_rows)
        public_right = public_matrix.slice(0, d, p1_rows, p1_rows + p2_rows)
        p_hat_image = (public_left * p1) + (public_right * p2)
        if p_hat_image.col_size() != target_cols:
            p_hat_image = p_hat_image.slice_columns(0, target_cols)
        perturbed = target - p_hat_image
        z_hat = perturbed.gauss_samp_gq_arb_base(self.c, self.sigma, random_gpu_rng_seed())
        r_z = td.r * z_hat
        out = GpuDCRTPolyMatrix(params, p1_rows + p2_rows, target_cols, p1.level, p1.is_ntt)
        out.copy_block_from(p1, 0, 0, 0, 0, p1_rows, target_cols)
        out.copy_block_from(p2, p1_rows, 0, 0, 0, p2_rows, target_cols)
        del p1, p2
        out.add_block_from(r_z, 0, 0, 0, 0, r_z.row_size(), target_cols)
        del r_z
        e_z = td.e * z_hat
        out.add_block_from(e_z, d, 0, 0, 0, e_z.row_size(), target_cols)
        del e_z
        out.add_block_from(z_hat, 2 * d, 0, 0, 0, z_hat.row_size(), target_cols)
        return out

    def preimage_extend(self, params, td, public_matrix, ext_matrix, target) -> GpuDCRTPolyMatrix:
        """gpu.rs:399-420."""
        d = public_matrix.row_size()
        n, k = params.ring_dimension(), params.modulus_digits()
        s = preimage_smoothing_parameter(self.base, self.sigma, d, n, k)
        u = GpuDCRTPolyUniformSampler()
        right = u.sample_uniform(params, ext_matrix.col_size(), target.col_size(), DistType.GaussDist(s))
        t = target - (ext_matrix * right)
        left = self.preimage(params, td, public_matrix, t)
        return left.concat_rows([right])

    # ---- several requests against one trapdoor in ONE sequence of launches ------------------------------------------
    BATCH_BYTES = 1 << 30  # cap on the perturbation block p2 of one batch (bounds the batch's device memory)

    @staticmethod
    def _segments_supported(params, td: GpuDCRTTrapdoor) -> bool:
        """what the library's *_segments entry points cover (include/gpupoly.h): n a multiple of 128, trapdoor dimension
        <= 2 (the p1 lane kernel), at most four digits per tower (the G-sampler's lane kernel), default RNG keying"""
        import os

        dpt = -(-params.crt_bits() // params.base_bits())
        return (params.ring_dimension() % 128 == 0 and td.r.row_size() <= 2 and dpt <= 4
                and not os.environ.get("MXX_HIP_RNG_COMPAT", "").startswith("r")
                and not os.environ.get("MXX_HIP_P1", "").startswith("s")
                and os.environ.get("MXX_PREIMAGE_BATCH", "1") != "0")

    def preimage_many(self, params, td: GpuDCRTTrapdoor, public_matrix, targets, _seeds=None) -> list:
        """[preimage(params, td, public_matrix, t) for t in targets] - the SAME matrices, bit for bit, for the same seeds -
        through one sequence of launches over the column-wise concatenation of the targets.  Every request keeps its own
        three seeds (drawn here request by request, in the order `preimage` draws them) and the samplers key every element by
        its position inside its own request (`gpupoly_*_segments`), so batching changes nothing a caller can observe except
        the time: at n = 256 a four-column request leaves the chip ~97 % idle and lasts as long as its unluckiest lane's
        chain of Karney steps, and sixteen of them last about as long as one.  Requests whose operands already fill the
        device (`_traffic_bound`) and shapes the segmented samplers do not cover go through `preimage` one by one."""
        targets = list(targets)
        seeds = list(_seeds) if _seeds is not None else [self._draw_seeds() for _ in targets]
        results = [None] * len(targets)
        dd, dk = td.r.row_size(), td.r.col_size()
        word = 4 if max(params.moduli()) < (1 << 31) else 8
        poly_bytes = params.crt_depth() * params.ring_dimension() * word
        batchable = []
        for j, t in enumerate(targets):
            pad = -(-t.col_size() // dd) * dd
            if t.col_size() == 0 or _traffic_bound(params, dk * pad) or not self._segments_supported(params, td):
                results[j] = self.preimage(params, td, public_matrix, t, _seeds=seeds[j])
            else:
                batchable.append(j)
        while batchable:
            group, size = [], 0
            while batchable and len(group) < 64:  # RNG_MAX_SEGMENTS per launch
                j = batchable[0]
                pad = -(-targets[j].col_size() // dd) * dd
                if group and size + dk * pad * poly_bytes > self.BATCH_BYTES:
                    break
                group.append(batchable.pop(0))
                size += dk * pad * poly_bytes
            if len(group) == 1:
                results[group[0]] = self.preimage(params, td, public_matrix, targets[group[0]], _seeds=seeds[group[0]])
                continue
            for j, x in zip(group, self._preimage_segments(params, td, public_matrix, [targets[j] for j in group], [seeds[j] for j in group])):
                results[j] = x
        return results

    @staticmethod
    def _drop_padding(m, cols, pads):
        """the first cols[j] columns of every pads[j]-wide segment (d > 1 pads a request's perturbation to a multiple of d)"""
        if cols == pads:
            return m
        widths = []
        for c, p_ in zip(cols, pads):
            widths += [c, p_ - c]
        parts = m.split_columns(widths)
        return GpuDCRTPolyMatrix.concat_columns_of(parts[0::2])

    def _preimage_segments(self, params, td: GpuDCRTTrapdoor, public_matrix, targets, seeds) -> list:
        """the body of `preimage` (gpu.rs:228-369, :423-474) over [u_0 | u_1 | ...] with a seed triple per segment"""
        M = GpuDCRTPolyMatrix
        d = public_matrix.row_size()
        dd, dk = td.r.row_size(), td.r.col_size()
        cols = [t.col_size() for t in targets]
        pads = [-(-c // dd) * dd for c in cols]
        for t in targets:
            assert t.row_size() == d, "Target matrix should have the same number of rows as the public matrix"
        n, k = params.ring_dimension(), params.modulus_digits()
        s = preimage_smoothing_parameter(self.base, self.sigma, d, n, k)
        dgg_large_std = math.sqrt(s * s - self.c * self.c)
        p1_rows, p2_rows = td.re.row_size(), td.re.col_size()
        assert public_matrix.col_size() == p1_rows + p2_rows, "public matrix columns must match perturbation rows"
        left, right, _ = td.public_matrix_parts(public_matrix)
        gauss = DistType.GaussDist(dgg_large_std)
        p2_pad = M.sample_distribution_segments(params, dk, pads, gauss.as_ffi(), gauss.sigma, [sd[0] for sd in seeds])
        tp2 = td.re * p2_pad
        cache = td.p1_covariance_cache(self.c, s, self.sigma)
        p1_pad = M.sample_p1_full_cached_segments(cache, tp2, [sd[1] for sd in seeds], pads)
        p1, p2 = self._drop_padding(p1_pad, cols, pads), self._drop_padding(p2_pad, cols, pads)
        perturbed = M.concat_columns_of(targets) - ((left * p1) + (right * p2))
        z_hat = perturbed.gauss_samp_gq_arb_base_segments(self.c, self.sigma, [sd[2] for sd in seeds], cols)
        re_z = td.re * z_hat
        out = M(params, p1_rows + p2_rows, sum(cols), p1.level, True)
        out.add_rows_from(0, p1, re_z)
        out.add_rows_from(p1_rows, p2, z_hat)
        return out.split_columns(cols)

    def preimage_batched_sharded(self, requests):
        """`preimage_batched_sharded` (gpu.rs:371-397): requests = [(entry_idx, params, trapdoor, A, target)];
        every request runs on the device context its params name, and the contexts work concurrently - the
        reference fans out with rayon's `into_par_iter`; here one worker thread per distinct context issues that
        context's requests (the ABI calls release the GIL and never block the host, so the devices' streams fill in
        parallel), and within a context the requests that share a trapdoor and a public matrix are sampled TOGETHER
        (`preimage_many`): same outputs as one by one, one sequence of launches.  Results come back in request order,
        like rayon's collect."""
        from concurrent.futures import ThreadPoolExecutor

        groups: dict = {}
        for pos, req in enumerate(requests):
            groups.setdefault(req[1].ctx_raw().value, []).append((pos, req))
        results = [None] * len(requests)

        def run(items):
            # requests of one context that share a trapdoor and a public matrix (the GGH15 callers' case: dozens of targets
            # per key, src/lookup/ggh15/pubkey_gpu.rs:615-971) go through `preimage_many` - one sequence of launches; the
            # seeds are drawn request by request in the order given, exactly as the one-by-one loop would draw them
            drawn = {pos: self._draw_seeds() for pos, _ in items}
            by_key: dict = {}
            for pos, (idx, p, td, a, t) in items:
                by_key.setdefault((id(td), id(a)), []).append((pos, idx, p, td, a, t))
            groups_ = list(by_key.values())
            nworkers = min(preimage_workers(), len(groups_))
            if nworkers <= 1:
                for members in groups_:
                    _, _, p, td, a, _ = members[0]
                    outs = self.preimage_many(p, td, a, [m[5] for m in members], _seeds=[drawn[m[0]] for m in members])
                    for (pos, idx, *_), x in zip(members, outs):
                        results[pos] = (idx, x)
                return
            # Different trapdoors in ONE context: the reference runs them concurrently on per-(matrix, limb) streams
            # (rayon over the requests, cuda/src/matrix/MatrixUtils.cu:259-400); a context here has one compute stream, so
            # the key groups are dealt to worker contexts - the same device, a stream and an allocator of their own
            # (`worker_params`).  Worker 0 is the requests' own context; the others get device-to-device replicas of
            # (trapdoor, A), cached on the trapdoor, and copies of their targets; the preimages come back by
            # device-to-device copies ordered on both streams (`gpupoly_matrix_copy_to_context`).  A small launch-bound
            # request fills a few percent of the chip, so several in flight overlap on the device (1.5x here; the host's issue
            # rate bounds it).
            plan = []  # (worker, members, params, trapdoor, A, targets) with every input already in the worker's context
            for g_i, members in enumerate(groups_):
                w = g_i % nworkers
                _, _, p, td, a, _ = members[0]
                if w == 0:
                    plan.append((w, members, p, td, a, [m[5] for m in members]))
                else:
                    pw = worker_params(p, w)
                    tdw, aw = td.replica_for(pw, a)
                    plan.append((w, members, pw, tdw, aw, [m[5].to_params(pw) for m in members]))
            # ONE host thread issues the groups, dealt round-robin to the worker contexts: no ABI call on this path waits
            # for the device, so the streams fill side by side.  (A host thread per worker was measured too: the Python
            # threads pass the interpreter lock back and forth - 3.21 ms against 2.98 from one thread, 4.49 on one stream,
            # for 8 keys x 2 requests on the M4 ring, tools/time_mixed_keys.py.  A Rust host has no such lock.)
            outs_by_group = [self.preimage_many(pw, tdw, aw, targets, _seeds=[drawn[m[0]] for m in members])
                             for _, members, pw, tdw, aw, targets in plan]
            for (w, members, *_), outs in zip(plan, outs_by_group):
                p = members[0][2]
                for (pos, idx, *_), x in zip(members, outs):
                    results[pos] = (idx, x if w == 0 else x.to_params(p))

        if len(groups) <= 1:
            for items in groups.values():
                run(items)
            return results
        with ThreadPoolExecutor(max_workers=len(groups)) as pool:
            for f in [pool.submit(run, items) for items in groups.values()]:
                f.result()  # re-raises a worker's exception
        return results

    def preimage_column_sharded(self, comm, shards):
        """One preimage of a wide target over several device contexts: `shards` = [(params, trapdoor, A, target columns
        of that context), ...] in communicator order (the trapdoor and A replicated with `to_params`).  Every context
        samples its column block concurrently - a worker thread per context, as `preimage_batched_sharded` - and ONE
        all-gather through the C ABI (`gpupoly_matrix_all_gather_columns`: RCCL on the contexts' own streams) leaves the
        whole preimage on every device.  The reference's fan-out (gpu.rs:371-397) returns the blocks to the host
        instead; INTEGRATION.md shows the Rust form."""
        from concurrent.futures import ThreadPoolExecutor

        assert len(shards) == len(comm), "one shard per context of the communicator"

        def run(item):
            p, td, a, t = item
            if t.col_size() == 0:
                k = p.modulus_digits()
                return GpuDCRTPolyMatrix(p, a.row_size() * (k + 2), 0, p.crt_depth() - 1, True)
            return self.preimage(p, td, a, t)

        if len(shards) == 1:
            blocks = [run(shards[0])]
        else:
            with ThreadPoolExecutor(max_workers=len(shards)) as pool:
                blocks = list(pool.map(run, shards))
        return comm.all_gather_columns(blocks)

