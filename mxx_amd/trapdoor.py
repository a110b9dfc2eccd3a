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
from .sampler import DistType, GpuDCRTPolyUniformSampler, random_gpu_rng_seed

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
        per public-matrix OBJECT (callers keep a trapdoor with its matrix: `src/sampler/trapdoor/gpu.rs:228-369` slices A on
        every call) and reused while that object is alive; a caller that rewrites the matrix in place must pass a new one."""
        with self._p1_lock:
            if self._stacked is not None and self._stacked[0]() is public_matrix:
                return self._stacked[1]
            d, p1_rows, p2_rows = public_matrix.row_size(), self.re.row_size(), self.re.col_size()
            left = public_matrix.slice(0, d, 0, p1_rows)
            right = public_matrix.slice(0, d, p1_rows, p1_rows + p2_rows)
            parts = (left, right, self.re.concat_rows([right]))
            self._stacked = (weakref.ref(public_matrix), parts)
            return parts

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

    def _sample_pert_parts(self, params, td: GpuDCRTTrapdoor, s, c, dgg_stddev, sigma_large, total_ncol, stacked=None):
        """The same with the by-products the large-operand assembly reuses.  With `stacked` = [R; E; right] (right = the
        public matrix's columns over p2; `GpuDCRTTrapdoor.public_matrix_parts`) the product right * p2 the caller needs next
        rides in the same pass over p2 as [R;E] p2.  Returns p1, p2, [R;E] p2 (EVAL, kept for the final assembly) and
        right * p2 (or None); the last two are row views of the one product - no copies - and only the p1 sampler, which
        takes its argument to the coefficient domain in place, gets a slice of its own."""
        u = GpuDCRTPolyUniformSampler()
        d, dk = td.r.row_size(), td.r.col_size()
        padded = -(-total_ncol // d) * d
        p2 = u.sample_uniform(params, dk, padded, DistType.GaussDist(sigma_large))
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
        p1 = GpuDCRTPolyMatrix.sample_p1_full_cached(cache, tp2, random_gpu_rng_seed())
        return p1, p2, tp2_eval, rp2

    def preimage(self, params, td: GpuDCRTTrapdoor, public_matrix, target) -> GpuDCRTPolyMatrix:
        """x with public_matrix * x == target (gpu.rs:228-369)."""
        d = public_matrix.row_size()
        target_cols = target.col_size()
        assert target.row_size() == d, "Target matrix should have the same number of rows as the public matrix"
        n, k = params.ring_dimension(), params.modulus_digits()
        s = preimage_smoothing_parameter(self.base, self.sigma, d, n, k)
        dgg_large_std = math.sqrt(s * s - self.c * self.c)
        p1_rows, p2_rows = td.re.row_size(), td.re.col_size()
        assert public_matrix.col_size() == p1_rows + p2_rows, "public matrix columns must match perturbation rows"
        left, right, stacked = td.public_matrix_parts(public_matrix)
        p1, p2, tp2, rp2 = self._sample_pert_parts(params, td, s, self.c, self.sigma, dgg_large_std, target_cols, stacked)
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
            z = perturbed.gauss_samp_gq_arb_base(self.c, self.sigma, random_gpu_rng_seed(), coeff_out=True)
            out.ntt_add_rows_from(p1_rows, z, p2, consume=True)
            re_x = td.re * out.row_view(p1_rows, p1_rows + p2_rows)
            out.add_rows_from(0, p1 - tp2, re_x)
            return out
        z_hat = perturbed.gauss_samp_gq_arb_base(self.c, self.sigma, random_gpu_rng_seed())
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

    def preimage_batched_sharded(self, requests):
        """`preimage_batched_sharded` (gpu.rs:371-397): requests = [(entry_idx, params, trapdoor, A, target)];
        every request runs on the device context its params name, and the contexts work concurrently - the
        reference fans out with rayon's `into_par_iter`; here one worker thread per distinct context issues that
        context's requests in order (the ABI calls release the GIL and never block the host, so the devices'
        streams fill in parallel).  Results come back in request order, like rayon's collect."""
        from concurrent.futures import ThreadPoolExecutor

        groups: dict = {}
        for pos, req in enumerate(requests):
            groups.setdefault(req[1].ctx_raw().value, []).append((pos, req))
        results = [None] * len(requests)

        def run(items):
            for pos, (idx, p, td, a, t) in items:
                results[pos] = (idx, self.preimage(p, td, a, t))

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

