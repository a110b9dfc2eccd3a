// trapdoor.hip — G-lattice sampling and the p1 perturbation sampler of the
// trapdoor-preimage path.  Replaces cuda/src/matrix/MatrixTrapdoor.cu behind
// cuda/include/matrix/MatrixTrapdoor.cuh:66-101.
//
// G-sampling (Genise-Micciancio "GaussSampGqArbBase", the algorithm OpenFHE's
// DCRTGaussSampGqArbBase implements; reference kernel MatrixTrapdoor.cu:701-833):
// for tower t and coefficient value v in [0,q_t), sample z in Z^dpt from the coset
// of Lambda(g_t) with sum_d b^d z_d = v (mod q_t), width c.  One thread per
// (entry, tower, coefficient) samples the dpt digits and writes their residues
// into all output limbs directly (no int64 staging buffer, no scatter launch).
//
// p1 perturbation (reference MatrixTrapdoor.cu:95-360):
// per coefficient index the 2d x 2d covariance [[s^2 I - c^2 A, -c^2 B],
// [-c^2 B^T, s^2 I - c^2 D]] (A,B,D centred limb-0 coefficients) is factored once
// into conditional standard deviations + update columns (the cache); sampling then
// walks t = m-1..0 with Karney integers and mean updates.
#include "common.h"
#include "modarith.h"
#include "rng.h"

#include <algorithm>
#include <cstdlib>

static constexpr uint64_t kTagGadget = 0x6761646765746731ull;
static constexpr uint64_t kTagP1 = 0x7065727475726231ull;
static constexpr int kGaussMaxDigits = 64;

// ---- G-lattice sampler ------------------------------------------------------------------------
// The stream of element (polynomial p, tower t, coefficient i): sub-key (seed, kTagGadget, 0) - one per call -,
// stream0 = (i + 1) 2^8 + (t + 1) (at most 64 towers), stream1 = p + 1.
__host__ __device__ __forceinline__ uint64_t gadget_stream0(uint32_t i, uint32_t t) {
    return ((static_cast<uint64_t>(i) + 1) << 8) | (static_cast<uint64_t>(t) + 1);
}

template <typename W, int MAXD>
__global__ void __launch_bounds__(128) gauss_samp_gq_kernel(W *__restrict__ out, const W *__restrict__ src,
                                     const LimbConst *__restrict__ limbs, size_t src_polys, uint32_t src_cols,
                                     uint32_t L, uint32_t N, uint32_t dpt, uint32_t base_bits, double c, size_t k,
                                     GpuRngSeed seed) {
    __shared__ uint32_t ring[128 * RNG_RING_SLOTS];
    const size_t idx = item_index();
    const size_t total = src_polys * L * N;
    if (idx >= total) return;
    const uint32_t i = static_cast<uint32_t>(idx % N);
    const size_t pt = idx / N;
    const uint32_t t = static_cast<uint32_t>(pt % L);
    const size_t p = pt / L;
    const uint64_t qt = limbs[t].q;
    uint64_t value = static_cast<uint64_t>(src[(p * L + t) * N + i]) % qt;

    const uint64_t base = 1ull << base_bits;
    const double base_f = static_cast<double>(base);
    const double sigma = c / (base_f + 1.0);
    const double kf = static_cast<double>(dpt);

    int64_t m_digits[MAXD], v_digits[MAXD], z[MAXD];
    double a[MAXD], zf[MAXD], cvec[MAXD];
    {
        uint64_t mq = qt, vv = value;
#pragma unroll
        for (int d = 0; d < MAXD; ++d) {
            if (d < (int)dpt) {
                m_digits[d] = static_cast<int64_t>(mq % base);
                mq /= base;
                v_digits[d] = static_cast<int64_t>(vv % base);
                vv /= base;
            } else {
                m_digits[d] = 0;
                v_digits[d] = 0;
            }
        }
    }
    ChaChaRng rng;
    rng_init(rng, ring, seed, gadget_stream0(i, t), static_cast<uint64_t>(p) + 1, 0, kTagGadget);
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
        if ((d & 3) == 0) rng_fill_lane(rng);  // 4 normals = 32 draws per checkpoint
        zf[d] = d < (int)dpt ? sigma * rng_standard_normal(rng) : 0.0;
    }

    // perturbation p = L-factor applied to zf; l_d, h_d are the Cholesky entries of the
    // basis Gram matrix (Genise-Micciancio, alg. 3), folded into the running sums here
    {
        double prev_c = 0.0;
#pragma unroll
        for (int d = 0; d < MAXD; ++d) {
            if (d < (int)dpt) {
                cvec[d] = (prev_c + static_cast<double>(m_digits[d])) / base_f;
                prev_c = cvec[d];
            } else {
                cvec[d] = 0.0;
            }
        }
        double prev_a = 0.0;
#pragma unroll
        for (int d = 0; d < MAXD; ++d) {
            if (d < (int)dpt) {
                const double ld = d == 0 ? sqrt(base_f * (1.0 + 1.0 / kf) + 1.0)
                                         : sqrt(base_f * (1.0 + 1.0 / (kf - static_cast<double>(d))));
                double pd;
                if (d + 1 < (int)dpt) {
                    const double hn = sqrt(base_f * (1.0 - 1.0 / (kf - static_cast<double>(d))));
                    pd = ld * zf[d] + hn * zf[d + 1 < MAXD ? d + 1 : d];
                } else {
                    const double hd = d == 0 ? 0.0 : sqrt(base_f * (1.0 - 1.0 / (kf - static_cast<double>(d - 1))));
                    pd = hd * zf[d];
                }
                a[d] = (prev_a + static_cast<double>(v_digits[d]) - pd) / base_f;
                prev_a = a[d];
            } else {
                a[d] = 0.0;
            }
        }
    }
    const int last = static_cast<int>(dpt) - 1;
    double a_last = 0.0, c_last = 1.0;
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
        if (d == last) {
            a_last = a[d];
            c_last = cvec[d];
        }
    const int64_t z_last = sample_integer_karney(rng, -a_last / c_last, sigma / c_last);
#pragma unroll
    for (int d = 0; d < MAXD; ++d) a[d] += static_cast<double>(z_last) * cvec[d];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
        if (d < last) z[d] = sample_integer_karney(rng, -a[d], sigma);
        else z[d] = z_last;
    }

    const size_t r = p / src_cols, col = p - r * src_cols;
    int64_t z_prev = 0;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
        if (d < (int)dpt) {
            int64_t digit;
            if (dpt == 1) digit = static_cast<int64_t>(base) * z[0] + m_digits[0] * z[0] + v_digits[0];
            else if (d == 0) digit = static_cast<int64_t>(base) * z[0] + m_digits[0] * z_last + v_digits[0];
            else if (d < last) digit = static_cast<int64_t>(base) * z[d] - z_prev + m_digits[d] * z_last + v_digits[d];
            else digit = m_digits[d] * z_last - z_prev + v_digits[d];
            z_prev = z[d];
            const size_t orow = r * k + static_cast<size_t>(t) * dpt + d;
            const size_t opoly = orow * src_cols + col;
            for (uint32_t l = 0; l < L; ++l)
                out[(opoly * L + l) * N + i] = signed_to_residue_mu<W>(digit, limbs[l].q, limbs[l].mu64);
        }
    }
}

// ---- G-lattice sampler, persistent-lane form (dpt <= 4; rng.h explains the scheme) -------------
// per-tower constants of the sampler: c_last = c_{dpt-1} of the tower's modulus digits, the width
// sigma / c_last of the first Karney integer and its divisor
struct GqTower {
    double c_last, sd;
    KarneyDivisor div;
    double cvec[4];  // c_d of the tower's modulus digits (the lane kernel handles at most four digits per tower)
};

__global__ void gq_tower_kernel(GqTower *__restrict__ towers, const LimbConst *__restrict__ limbs, uint32_t L,
                                uint32_t dpt, uint32_t base_bits, double c) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const uint64_t base = 1ull << base_bits;
    const double base_f = static_cast<double>(base);
    const double sigma = c / (base_f + 1.0);
    const uint64_t qt = limbs[t].q;
    double c_last = 0.0;
    GqTower g;
    for (uint32_t d = 0; d < 4; ++d) g.cvec[d] = 0.0;
    for (uint32_t d = 0; d < dpt; ++d) {
        c_last = (c_last + static_cast<double>(static_cast<int64_t>((qt >> (base_bits * d)) & (base - 1)))) / base_f;
        if (d < 4) g.cvec[d] = c_last;
    }
    g.c_last = c_last;
    g.sd = sigma / c_last;
    g.div = karney_divisor(g.sd);
    towers[t] = g;
}

// Pass 1, fully convergent: the dpt normals every element draws first (draws 0..8*dpt-1 = 64-bit words 0..2*dpt-1 of
// keystream block 0) and the centres a_d they imply.  a_out is [dpt][total]; the 8 - 2*dpt 64-bit words of block 0 the
// normals did not use are handed to pass 2 in left_out ([8 - 2*dpt][total]).
// l_d / h_d of the perturbation's Cholesky-like recurrence (MatrixTrapdoor.cu:701-833): functions of (base, dpt, d) only,
// evaluated once on the host (IEEE sqrt and division: the same bits as the device's) instead of per element - four
// square roots and four divisions of the prep kernel's ~1475 instructions per element
struct GqPertConsts {
    double ld[4], hn[4];  // pd = ld[d] z_d + hn[d] z_{d+1} for d + 1 < dpt, hn[d - 1] z_d for the last digit (0 when dpt = 1)
};

static inline GqPertConsts gq_pert_consts(uint32_t dpt, uint32_t base_bits) {
    GqPertConsts k{};
    const double base_f = static_cast<double>(1ull << base_bits), kf = static_cast<double>(dpt);
    for (uint32_t d = 0; d < 4 && d < dpt; ++d) {
        k.ld[d] = d == 0 ? sqrt(base_f * (1.0 + 1.0 / kf) + 1.0) : sqrt(base_f * (1.0 + 1.0 / (kf - static_cast<double>(d))));
        k.hn[d] = sqrt(base_f * (1.0 - 1.0 / (kf - static_cast<double>(d))));
    }
    return k;
}

// SEG: the source is the column-wise concatenation of independently seeded requests (rng.h, RngSegments; the launcher
// checks that n is a multiple of 64, so a wave's elements share one polynomial): the segment's sub-key replaces `key`
// and the stream is keyed by the polynomial's index INSIDE its segment, row * segment columns + local column.
template <typename W, int MAXD, bool SEG = false>
__global__ void __launch_bounds__(256) gauss_samp_prep_kernel(double *__restrict__ a_out, uint64_t *__restrict__ left_out,
                                       const W *__restrict__ src, const LimbConst *__restrict__ limbs,
                                       ChaChaKey key, size_t total, uint32_t L, uint32_t logN,
                                       uint32_t dpt, uint32_t base_bits, double c, GqPertConsts pc,
                                       uint32_t src_cols, SegArg<SEG> segs) {
    static_assert(MAXD <= 4, "GqPertConsts holds four digits");
    const size_t idx = item_index();
    if (idx >= total) return;
    const uint32_t i = static_cast<uint32_t>(idx & ((1u << logN) - 1));
    const uint32_t pt = static_cast<uint32_t>(idx >> logN);
    uint32_t p = pt / L;
    const uint32_t t = pt - p * L;
    if constexpr (SEG) {
        const uint32_t p_u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(p));
        const uint32_t row = p_u / src_cols, col = p_u - row * src_cols;
        const uint32_t j = rng_segment_of(segs, col);
        key = segs.key[j];
        p = row * (segs.start[j + 1] - segs.start[j]) + (col - segs.start[j]);
    }
    const uint64_t qt = limbs[t].q;
    uint64_t value = static_cast<uint64_t>(src[idx]);
    if (value >= qt) value %= qt;
    const uint64_t base = 1ull << base_bits;
    const double base_f = static_cast<double>(base);
    const double sigma = c / (base_f + 1.0);

    uint64_t w[8];
    chacha_block_words(key, gadget_stream0(i, t), static_cast<uint64_t>(p) + 1, 0, w);
    double zf[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
        zf[d] = d < (int)dpt ? sigma * (sqrt(-2.0 * det_log(u64_to_open01(w[2 * d]))) * det_cos2pi(u64_to_open01(w[2 * d + 1]))) : 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (j >= 2 * (int)dpt) left_out[static_cast<size_t>(j - 2 * (int)dpt) * total + idx] = w[j];
    double prev_a = 0.0;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) {
        if (d < (int)dpt) {
            double pd;
            if (d + 1 < (int)dpt) pd = pc.ld[d] * zf[d] + pc.hn[d] * zf[d + 1 < MAXD ? d + 1 : d];
            else pd = (d == 0 ? 0.0 : pc.hn[d - 1 < 0 ? 0 : d - 1]) * zf[d];
            const double vd = static_cast<double>(static_cast<int64_t>((value >> (base_bits * d)) & (base - 1)));
            const double ad = (prev_a + vd - pd) / base_f;
            prev_a = ad;
            a_out[static_cast<size_t>(d) * total + idx] = ad;
        }
    }
}

// refill cadence of pass 2 (fill_every, in checkpoints): an element starts with the 32 - 8 * dpt draws block 0 has left
// over.  With 16 of them (dpt <= 2) that is enough for the first checkpoint or two, so refills are held to every
// third checkpoint (same-box A/B on M3A, lanes kernel: every checkpoint 3.33 ms, every 2nd 2.78, every 3rd 2.63,
// every 4th 2.74; profiles/r03_notes.md); with fewer or none left (dpt = 3, 4) the longer elements still make every
// third the best cadence (tools/time_gsamp.py: 0.572 / 0.537 / 0.527 / 0.533 ms at dpt = 3, 0.686 / 0.666 / 0.640 / 0.657 at 4).
// Pass 2: the dpt Karney integers of every element.  Wave w owns elements [w*64*per_lane,
// +64*per_lane) and its lanes take them one at a time (wave_take, rng.h).
// ph = index of the integer in flight (0: z_last, 1+d: z_d).
// UNI: every wave's chunk lies inside ONE (polynomial, tower) vector (the launcher checks that 64 * per_lane divides n):
// the tower's constants, the polynomial index and the division that finds them are then per-WAVE scalars, read once
// with scalar loads, instead of a 32-bit division and eleven vector loads with their waits in every element hand-over
// (the second stall, c_d of the tower at the first integer's hand-over, goes with them).
// SEG (with UNI): independently seeded column segments, as in gauss_samp_prep_kernel - the chunk's polynomial lies in one
// segment; its sub-key and the polynomial's index inside the segment are per-wave scalars.
template <typename W, int MAXD, int SV, bool UNI = false, bool SEG = false>
__global__ void __launch_bounds__(SAMPLER_THREADS, 4) gauss_samp_lanes_kernel(int64_t *__restrict__ stage,
                                        const LimbConst *__restrict__ limbs, ChaChaKey key,
                                        const GqTower *__restrict__ towers, const double *__restrict__ a_in,
                                        const uint64_t *__restrict__ left_in, size_t total, uint32_t src_cols,
                                        uint32_t L, uint32_t logN, uint32_t dpt, uint32_t base_bits, double c,
                                        KarneyDivisor div_sigma, uint32_t per_lane, uint32_t fill_every,
                                        SegArg<SEG> segs) {
    static_assert(!SEG || UNI, "segments need a wave's chunk inside one polynomial");
    __shared__ uint32_t ring[SAMPLER_THREADS * RNG_RING_SLOTS];  // 128 bytes per lane: 8 KB per one-wave workgroup
    const uint64_t base = 1ull << base_bits;
    const double base_f = static_cast<double>(base);
    const double sigma = c / (base_f + 1.0);
    const int last = static_cast<int>(dpt) - 1;
    const uint32_t nleft = 8 - 2 * dpt;
    WaveChunk chunk = wave_chunk(total, per_lane);
    // UNI: (p, t) of the whole chunk and the tower's record, wave-uniform
    const uint32_t pt_u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(chunk.base >> logN)));
    uint32_t p_u = pt_u / L;
    const uint32_t t_u = pt_u - p_u * L;
    const GqTower tw_u = towers[UNI ? t_u : 0];
    if constexpr (SEG) {
        const uint32_t row = p_u / src_cols, col = p_u - row * src_cols;
        const uint32_t j = rng_segment_of(segs, col);
        key = segs.key[j];
        p_u = row * (segs.start[j + 1] - segs.start[j]) + (col - segs.start[j]);
    }

    ChaChaRng rng;
    rng_init_keyed(rng, ring, key, 0, 0, SAMPLER_THREADS);
    KarneyFsm f;
    karney_reset(f);
    bool fin = true, have = false;
    uint32_t ph = 0;
    size_t idx = 0;
    uint32_t tower = 0;
    double a[MAXD];
    int64_t z[MAXD];
    int64_t z_last = 0;
#pragma unroll
    for (int d = 0; d < MAXD; ++d) { a[d] = 0.0; z[d] = 0; }

    // a superstep = 8 steps: checkpoint, then twice [finished integer handed on, service point, four cheap steps].
    // Written out instead of `if ((step & 3) == 0)` inside a one-step loop: the compiler then keeps the state machine
    // in place instead of copying ~30 registers around the branch on every step (sample_gauss_kernel: 2.42 -> 2.26 ms).
    auto integer_ready = [&]() {
        if (f.st == KS_DONE && !fin) {  // an integer of the current element is ready
            if (ph == 0) {
                z_last = f.result;
                const double zl = static_cast<double>(z_last);
#pragma unroll
                for (int d = 0; d < MAXD; ++d)
                    if (d < (int)dpt) a[d] += zl * (UNI ? tw_u.cvec[d] : towers[tower].cvec[d]);  // c_d: per tower, from gq_tower_kernel
            } else {
#pragma unroll
                for (int d = 0; d < MAXD; ++d)
                    if (d == (int)ph - 1) z[d] = f.result;
            }
            ++ph;
            if (ph == dpt) {
                fin = true;
            } else {
                double an = 0.0;
#pragma unroll
                for (int d = 0; d < MAXD; ++d)
                    if (d == (int)ph - 1) an = a[d];
                karney_begin(f, -an, sigma, div_sigma);
            }
        }
    };
    auto element_done = [&]() {  // element complete: write its digits, take the next one
        const bool take = f.st == KS_DONE && fin;  // element complete: write its digits, take the next one
        if (take && have) {
#pragma unroll
            for (int d = 0; d < MAXD; ++d) {
                if (d < (int)dpt) {
                    stage[idx * dpt + d] = d < last ? z[d] : z_last;  // pass 3 turns the integers into digits
                }
            }
        }
        const uint32_t e = wave_take(chunk, take);
        if (take) {
            have = e < chunk.len;
            if (have) {
                idx = chunk.base + e;
                const uint32_t i = static_cast<uint32_t>(idx & ((1u << logN) - 1));
                uint32_t p, t;
                if constexpr (UNI) {
                    p = p_u;
                    t = t_u;
                } else {
                    const uint32_t pt = static_cast<uint32_t>(idx >> logN);
                    p = pt / L;
                    t = pt - p * L;
                }
                tower = t;
                // block 0 went to pass 1: its unused words are draws 8*dpt..31 of the stream, continue at block 1
                rng_reopen(rng, gadget_stream0(i, t), static_cast<uint64_t>(p) + 1, 1);
                const uint32_t slot0 = (rng.tail >> RNG_DRAW_LOG) & (RNG_RING_SLOTS - 1);
                for (uint32_t w = 0; w < nleft; ++w) {
                    const uint64_t lw = left_in[static_cast<size_t>(w) * total + idx];
                    rng.ring[(slot0 + 4 * dpt + 2 * w) * rng.ring_stride] = static_cast<uint32_t>(lw);
                    rng.ring[(slot0 + 4 * dpt + 2 * w + 1) * rng.ring_stride] = static_cast<uint32_t>(lw >> 32);
                }
                rng.head = rng.tail + 2 * RNG_U64_DRAWS * dpt;
                rng.tail += RNG_BLOCK_DRAWS;
                double a_last = 0.0;
#pragma unroll
                for (int d = 0; d < MAXD; ++d) {
                    if (d < (int)dpt) {
                        a[d] = a_in[static_cast<size_t>(d) * total + idx];
                        a_last = a[d];
                    }
                }
                ph = 0;
                fin = false;
                if constexpr (UNI) {
                    karney_begin(f, -a_last / tw_u.c_last, tw_u.sd, tw_u.div);
                } else {
                    const GqTower tw = towers[t];
                    karney_begin(f, -a_last / tw.c_last, tw.sd, tw.div);
                }
            } else {
                f.st = KS_IDLE;
            }
        }
    };
    for (uint32_t step = 0;; ++step) {  // one superstep per iteration
        integer_ready();
        element_done();
        if (__all(f.st == KS_IDLE)) break;
        rng_fill_wave(rng, f.st != KS_IDLE, step % fill_every == 0, 16, karney_urgent(SV));
        karney_heavy(f, rng);
#pragma unroll
        for (int s4 = 0; s4 < KARNEY_SUPERSTEP / SV; ++s4) karney_light(f, rng);
#pragma unroll
        for (int sv = 1; sv < SV; ++sv) {
            integer_ready();
            karney_heavy(f, rng);
#pragma unroll
            for (int s4 = 0; s4 < KARNEY_SUPERSTEP / SV; ++s4) karney_light(f, rng);
        }
    }
}

// Pass 3, coalesced: the dpt integers of element (p, t, i) -> its digits -> residues in every limb of output rows
// r*k + t*dpt + d.  (Pass 2's lanes finish at different times; storing L residues per digit from there made every
// store a lone 32-byte HBM write - 6x the algorithmic bytes.  The digit arithmetic lives here too: this pass is
// bound by its stores, pass 2 by its instruction count - lanes kernel 2.56 -> 2.47 ms.)
template <typename W>
__global__ void __launch_bounds__(256) gauss_samp_expand_kernel(W *__restrict__ out, const int64_t *__restrict__ stage,
                                         const W *__restrict__ src, const LimbConst *__restrict__ limbs, size_t total,
                                         uint32_t src_cols, uint32_t L, uint32_t logN, uint32_t dpt, uint32_t k,
                                         uint32_t base_bits) {
    const size_t idx = item_index();
    if (idx >= total) return;
    const uint32_t i = static_cast<uint32_t>(idx & ((1u << logN) - 1));
    const uint32_t pt = static_cast<uint32_t>(idx >> logN);
    const uint32_t p = pt / L, t = pt - p * L;
    const uint32_t r = p / src_cols, col = p - r * src_cols;
    const uint64_t qt = limbs[t].q, base = 1ull << base_bits;
    uint64_t value = static_cast<uint64_t>(src[idx]);
    if (value >= qt) value %= qt;
    const int last = static_cast<int>(dpt) - 1;
    const int64_t z_last = stage[idx * dpt + last];
    int64_t z_prev = 0;
    for (uint32_t d = 0; d < dpt; ++d) {
        const int64_t md = static_cast<int64_t>((qt >> (base_bits * d)) & (base - 1));
        const int64_t vd = static_cast<int64_t>((value >> (base_bits * d)) & (base - 1));
        const int64_t zd = stage[idx * dpt + d];
        int64_t digit;
        if (dpt == 1) digit = static_cast<int64_t>(base) * zd + md * zd + vd;
        else if (d == 0) digit = static_cast<int64_t>(base) * zd + md * z_last + vd;
        else if (static_cast<int>(d) < last) digit = static_cast<int64_t>(base) * zd - z_prev + md * z_last + vd;
        else digit = md * z_last - z_prev + vd;
        z_prev = zd;
        const size_t orow = static_cast<size_t>(r) * k + t * dpt + d;
        W *dst = out + (((orow * src_cols + col) * L) << logN) + i;
        for (uint32_t l = 0; l < L; ++l)
            dst[static_cast<size_t>(l) << logN] = signed_to_residue_mu<W>(digit, limbs[l].q, limbs[l].mu64);
    }
}

// segs != nullptr: the source's columns are independently seeded segments (gpupoly_matrix_gauss_samp_gq_arb_base_segments);
// `seed` is then unused and the lane kernel runs in its UNI form with a lane count that keeps it there
template <typename W, int MAXD>
static int launch_gauss_samp_lanes(GpuContext *ctx, W *out, const W *src, size_t total, uint32_t src_cols, uint32_t L,
                                   uint32_t dpt, uint32_t base_bits, double c, size_t k, GpuRngSeed seed,
                                   const RngSegments *segs = nullptr) {
    if ((total >> ctx->logN) >> 32 || k >> 32) return set_error("gpu_matrix_gauss_samp_gq_arb_base: matrix too large");
    // [L] towers | [8][total] words between passes 1 and 2 | [total][dpt] digits
    void *towers = nullptr, *a_buf = nullptr, *stage = nullptr;
    if (ctx_alloc(ctx, static_cast<size_t>(L) * sizeof(GqTower), &towers) ||
        ctx_alloc(ctx, total * 8 * sizeof(uint64_t), &a_buf) || ctx_alloc(ctx, total * dpt * sizeof(int64_t), &stage)) {
        ctx_free(ctx, towers);
        ctx_free(ctx, a_buf);
        return 1;
    }
    double *a_words = static_cast<double *>(a_buf);
    uint64_t *left_words = static_cast<uint64_t *>(a_buf) + total * dpt;
    const ChaChaKey key = chacha_subkey(seed, 0, kTagGadget);
    MXX_LAUNCH(gq_tower_kernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<GqTower *>(towers), ctx->d_limbs,
                       L, dpt, base_bits, c);
    // per element: the residue read; dpt centres + the 8 - 2 dpt unused keystream words handed to pass 2
    MXX_TRACE_BYTES(static_cast<double>(total) * (sizeof(W) + 8.0 * (8 - dpt)));
    if (segs)
        MXX_LAUNCH((gauss_samp_prep_kernel<W, MAXD, true>), item_grid(total, 256), dim3(256), 0,
                           ctx->stream, a_words, left_words, src, ctx->d_limbs, key, total, L,
                           ctx->logN, dpt, base_bits, c, gq_pert_consts(dpt, base_bits), src_cols, *segs);
    else
        MXX_LAUNCH((gauss_samp_prep_kernel<W, MAXD>), item_grid(total, 256), dim3(256), 0,
                           ctx->stream, a_words, left_words, src, ctx->d_limbs, key, total, L,
                           ctx->logN, dpt, base_bits, c, gq_pert_consts(dpt, base_bits), src_cols, NoSegments{});
    uint32_t per_lane =
        sampler_per_lane(total, reinterpret_cast<const void *>(gauss_samp_lanes_kernel<W, MAXD, KARNEY_SERVICES>), ctx->device, ctx->env.sampler_per_lane);
    if (segs)  // a divisor of n / 64: every wave's chunk inside one (polynomial, tower) vector, i.e. inside one segment
        while ((static_cast<size_t>(ctx->N) / SAMPLER_THREADS) % per_lane) --per_lane;
    const unsigned blocks = static_cast<unsigned>((total + SAMPLER_THREADS * per_lane - 1) / (SAMPLER_THREADS * per_lane));
    const double sigma = c / (static_cast<double>(1ull << base_bits) + 1.0);
    MXX_TRACE_BYTES(static_cast<double>(total) * (8.0 * (8 - dpt) + 8.0 * dpt));  // pass 1's words read, dpt int64 digits written
    // a chunk of 64 * per_lane consecutive elements inside one (polynomial, tower) vector: the kernel's UNI form
    const bool uni = (static_cast<size_t>(ctx->N) % (static_cast<size_t>(SAMPLER_THREADS) * per_lane)) == 0 && !ctx->env.gsamp_no_uni;
#define LAUNCH_GL(SV, UNI)                                                                                                  \
    MXX_LAUNCH((gauss_samp_lanes_kernel<W, MAXD, SV, UNI>), dim3(blocks), dim3(SAMPLER_THREADS), 0, ctx->stream,            \
               static_cast<int64_t *>(stage), ctx->d_limbs, key, static_cast<const GqTower *>(towers), a_words, left_words, \
               total, src_cols, L, ctx->logN, dpt, base_bits, c, karney_divisor(sigma), per_lane,                           \
               static_cast<uint32_t>(ctx->env.sampler_fill_every ? ctx->env.sampler_fill_every : 3), NoSegments{})
#define LAUNCH_GLS(SV)                                                                                                      \
    MXX_LAUNCH((gauss_samp_lanes_kernel<W, MAXD, SV, true, true>), dim3(blocks), dim3(SAMPLER_THREADS), 0, ctx->stream,     \
               static_cast<int64_t *>(stage), ctx->d_limbs, key, static_cast<const GqTower *>(towers), a_words, left_words, \
               total, src_cols, L, ctx->logN, dpt, base_bits, c, karney_divisor(sigma), per_lane,                           \
               static_cast<uint32_t>(ctx->env.sampler_fill_every ? ctx->env.sampler_fill_every : 3), *segs)
    if (segs) {
        if (per_lane == 1) LAUNCH_GLS(1);
        else LAUNCH_GLS(KARNEY_SERVICES);
    } else if (per_lane == 1) {
        if (uni) LAUNCH_GL(1, true);
        else LAUNCH_GL(1, false);
    } else {
        if (uni) LAUNCH_GL(KARNEY_SERVICES, true);
        else LAUNCH_GL(KARNEY_SERVICES, false);
    }
#undef LAUNCH_GLS
#undef LAUNCH_GL
    // the int64 digits and the residue read; every digit written as a residue of every limb (the call's output: 655 MB at M3A)
    MXX_TRACE_BYTES(static_cast<double>(total) * (8.0 * dpt + sizeof(W) + static_cast<double>(dpt) * L * sizeof(W)));
    MXX_LAUNCH(gauss_samp_expand_kernel<W>, item_grid(total, 256), dim3(256), 0,
                       ctx->stream, out, static_cast<const int64_t *>(stage), src, ctx->d_limbs, total, src_cols, L, ctx->logN,
                       dpt, static_cast<uint32_t>(k), base_bits);
    const hipError_t err = hipGetLastError();
    ctx_free(ctx, towers);
    ctx_free(ctx, a_buf);
    ctx_free(ctx, stage);
    HIP_TRY(err);
    return 0;
}

template <typename W>
static int launch_gauss_samp(GpuContext *ctx, W *out, const W *src, size_t polys, uint32_t src_cols, uint32_t L,
                             uint32_t dpt, uint32_t base_bits, double c, size_t k, GpuRngSeed seed,
                             const RngSegments *segs = nullptr) {
    const size_t total = polys * L * static_cast<size_t>(ctx->N);
    // MXX_HIP_GSAMP=simple keeps the one-thread-per-element kernel for every dpt (A/B runs, tests)
    const bool simple = ctx->env.gsamp_simple && !segs;
    if (!simple && dpt <= 2) return launch_gauss_samp_lanes<W, 2>(ctx, out, src, total, src_cols, L, dpt, base_bits, c, k, seed, segs);
    if (!simple && dpt <= 4) return launch_gauss_samp_lanes<W, 4>(ctx, out, src, total, src_cols, L, dpt, base_bits, c, k, seed, segs);
    if (segs) return set_error("gpupoly_matrix_gauss_samp_gq_arb_base_segments: unsupported: more than four digits per tower");
    const dim3 blocks = item_grid(total, 128);
    const uint32_t N = static_cast<uint32_t>(ctx->N);
    MXX_TRACE_BYTES(static_cast<double>(total) * (sizeof(W) + static_cast<double>(dpt) * L * sizeof(W)));
#define LAUNCH_GS(MAXD)                                                                                        \
    MXX_LAUNCH((gauss_samp_gq_kernel<W, MAXD>), dim3(blocks), dim3(128), 0, ctx->stream, out, src,      \
                       ctx->d_limbs, polys, src_cols, L, N, dpt, base_bits, c, k, seed)
    if (dpt <= 2) LAUNCH_GS(2);
    else if (dpt <= 4) LAUNCH_GS(4);
    else if (dpt <= 8) LAUNCH_GS(8);
    else if (dpt <= 20) LAUNCH_GS(20);
    else LAUNCH_GS(64);
#undef LAUNCH_GS
    HIP_TRY(hipGetLastError());
    return 0;
}

static int gauss_samp_impl(GpuMatrix *src, uint32_t base_bits, double c, GpuRngSeed seed, GpuMatrix *out, const RngSegments *segs);

extern "C" int gpu_matrix_gauss_samp_gq_arb_base(GpuMatrix *src, uint32_t base_bits, double c, double dgg_stddev,
                                                 GpuRngSeed seed, GpuMatrix *out) {
    ABI_GUARD_BEGIN
    (void)dgg_stddev;  // unused by the reference too (MatrixTrapdoor.cu:1680)
    return gauss_samp_impl(src, base_bits, c, seed, out, nullptr);
    ABI_GUARD_END
}

// The same over independently seeded column segments (rng.h, RngSegments): columns [sum seg_cols[<j], + seg_cols[j]) of
// `out` equal what the plain entry point writes for the rows x seg_cols[j] source made of those columns under seeds[j].
extern "C" int gpupoly_matrix_gauss_samp_gq_arb_base_segments(GpuMatrix *src, uint32_t base_bits, double c, double dgg_stddev,
                                                              const GpuRngSeed *seeds, const size_t *seg_cols, size_t nseg,
                                                              GpuMatrix *out) {
    ABI_GUARD_BEGIN
    (void)dgg_stddev;
    if (!src || !out) return set_error("invalid gpupoly_matrix_gauss_samp_gq_arb_base_segments arguments");
    if (static_cast<size_t>(src->ctx->N) % SAMPLER_THREADS)
        return set_error("gpupoly_matrix_gauss_samp_gq_arb_base_segments: unsupported: ring too small for a wave per polynomial chunk");
    RngSegments segs;
    if (!rng_segments_build(segs, seeds, seg_cols, nseg, kTagGadget, src->cols))
        return set_error("gpupoly_matrix_gauss_samp_gq_arb_base_segments: segments must be 1..64, non-empty and cover the source's columns");
    return gauss_samp_impl(src, base_bits, c, GpuRngSeed{}, out, &segs);
    ABI_GUARD_END
}

static int gauss_samp_impl(GpuMatrix *src, uint32_t base_bits, double c, GpuRngSeed seed, GpuMatrix *out, const RngSegments *segs) {
    {
    if (!src || !out) return set_error("invalid gpu_matrix_gauss_samp_gq_arb_base arguments");
    if (base_bits == 0 || base_bits >= 63) return set_error("invalid base_bits in gpu_matrix_gauss_samp_gq_arb_base");
    if (!(c > 0.0)) return set_error("c must be positive in gpu_matrix_gauss_samp_gq_arb_base");
    if (src->ctx != out->ctx || src->level != out->level)
        return set_error("context mismatch in gpu_matrix_gauss_samp_gq_arb_base");
    GpuContext *ctx = src->ctx;
    const int requested = out->format;
    const size_t L = matrix_limbs(src);
    const uint32_t dpt = (ctx->crt_bits + base_bits - 1) / base_bits;
    if (dpt == 0 || dpt > kGaussMaxDigits)
        return set_error("invalid digits_per_tower in gpu_matrix_gauss_samp_gq_arb_base");
    const size_t k = static_cast<size_t>(dpt) * L;
    if (out->rows != src->rows * k || out->cols != src->cols)
        return set_error("output size mismatch in gpu_matrix_gauss_samp_gq_arb_base");
    const size_t polys = matrix_polys(src);
    if (polys == 0) {
        out->format = GPU_POLY_FORMAT_EVAL;
        return 0;
    }
    if (ctx_activate(ctx)) return 1;
    if (src->format == GPU_POLY_FORMAT_EVAL) {  // the source is consumed: convert in place
        int rc = gpu_matrix_intt_all(src);
        if (rc) return rc;
    }
    int rc = ctx->wide ? launch_gauss_samp<uint64_t>(ctx, static_cast<uint64_t *>(out->data),
                                                     static_cast<const uint64_t *>(src->data), polys,
                                                     (uint32_t)src->cols, (uint32_t)L, dpt, base_bits, c, k, seed, segs)
                       : launch_gauss_samp<uint32_t>(ctx, static_cast<uint32_t *>(out->data),
                                                     static_cast<const uint32_t *>(src->data), polys,
                                                     (uint32_t)src->cols, (uint32_t)L, dpt, base_bits, c, k, seed, segs);
    if (rc) return rc;
    out->format = GPU_POLY_FORMAT_COEFF;
    if (requested == GPU_POLY_FORMAT_EVAL) {
        rc = launch_ntt(ctx, out->data, matrix_polys(out) * L, static_cast<int>(L), false);
        if (rc) return rc;
        out->format = GPU_POLY_FORMAT_EVAL;
    }
    return 0;
    }
}

// ---- p1 perturbation sampler -------------------------------------------------------------------
template <typename W>
__global__ void p1_covariance_kernel(const W *__restrict__ a_mat, const W *__restrict__ b_mat,
                                     const W *__restrict__ d_mat, uint32_t d, uint32_t L, uint32_t N, uint64_t q0,
                                     double sigma, double s, double dgg_stddev, double *__restrict__ cov_ws,
                                     double *__restrict__ sqrt_var_out, double *__restrict__ update_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t m = 2 * d;
    double *cov = cov_ws + static_cast<size_t>(i) * m * m;
    double *sqrt_var = sqrt_var_out + static_cast<size_t>(i) * m;
    double *upd = update_out + static_cast<size_t>(i) * m * m;
    const double sigma2 = sigma * sigma, s2 = s * s, fallback = dgg_stddev * dgg_stddev, eps = 1e-9;
    // limb 0 of entry (r,c): word offset ((r*d+c)*L + 0)*N + i
    for (uint32_t r = 0; r < d; ++r)
        for (uint32_t c = 0; c < d; ++c) {
            const size_t rc = (static_cast<size_t>(r) * d + c) * L * N + i;
            const size_t cr = (static_cast<size_t>(c) * d + r) * L * N + i;
            const double a_rc = static_cast<double>(centered_residue(a_mat[rc], q0));
            const double d_rc = static_cast<double>(centered_residue(d_mat[rc], q0));
            const double b_rc = static_cast<double>(centered_residue(b_mat[rc], q0));
            const double b_cr = static_cast<double>(centered_residue(b_mat[cr], q0));
            cov[r * m + c] = -sigma2 * a_rc + (r == c ? s2 : 0.0);
            cov[(r + d) * m + (c + d)] = -sigma2 * d_rc + (r == c ? s2 : 0.0);
            cov[r * m + (c + d)] = -sigma2 * b_rc;
            cov[(r + d) * m + c] = -sigma2 * b_cr;
        }
    for (int t = static_cast<int>(m) - 1; t >= 0; --t) {
        double var = cov[t * m + t];
        if (!(var > eps)) var = fallback;
        sqrt_var[t] = sqrt(var);
        for (int r = 0; r < t; ++r) upd[t * m + r] = cov[r * m + t] / var;
        if (t == 0) break;
        for (int r = 0; r < t; ++r) {
            const double cr = upd[t * m + r];
            for (int c = 0; c <= r; ++c) {
                const double colc = upd[t * m + c] * var;
                const double v = cov[r * m + c] - cr * colc;
                cov[r * m + c] = v;
                cov[c * m + r] = v;
            }
        }
    }
}

// MAXM > 0: mean/sample vectors in registers; MAXM == 0: per-thread slices of a global workspace
template <typename W, int MAXM>
__global__ void __launch_bounds__(128) p1_sample_kernel(W *__restrict__ out, const W *__restrict__ tp2,
                                 const LimbConst *__restrict__ limbs, const double *__restrict__ sqrt_var_base,
                                 const double *__restrict__ update_base, uint32_t m, uint32_t cols, uint32_t L,
                                 uint32_t N, uint64_t q0, double c_scale, GpuRngSeed seed,
                                 double *__restrict__ mean_ws) {
    __shared__ uint32_t ring[128 * RNG_RING_SLOTS];
    const size_t idx = item_index();
    if (idx >= static_cast<size_t>(cols) * N) return;
    const uint32_t col = static_cast<uint32_t>(idx / N);
    const uint32_t i = static_cast<uint32_t>(idx - static_cast<size_t>(col) * N);
    const double *sqrt_var = sqrt_var_base + static_cast<size_t>(i) * m;
    const double *upd = update_base + static_cast<size_t>(i) * m * m;
    ChaChaRng rng;
    rng_init(rng, ring, seed, static_cast<uint64_t>(col) + 1, static_cast<uint64_t>(i) + 1, 0, kTagP1);

    if constexpr (MAXM > 0) {
        double mean[MAXM];
#pragma unroll
        for (int r = 0; r < MAXM; ++r)
            mean[r] = r < (int)m ? c_scale * static_cast<double>(centered_residue(
                                                 tp2[((static_cast<size_t>(r) * cols + col) * L) * N + i], q0))
                                 : 0.0;
#pragma unroll
        for (int t = MAXM - 1; t >= 0; --t) {
            if (t >= (int)m) continue;
            const double mu = mean[t];
            const int64_t z = sample_integer_karney(rng, mu, sqrt_var[t]);
            for (uint32_t l = 0; l < L; ++l)
                out[((static_cast<size_t>(t) * cols + col) * L + l) * N + i] =
                    signed_to_residue_mu<W>(z, limbs[l].q, limbs[l].mu64);
            const double delta = static_cast<double>(z) - mu;
#pragma unroll
            for (int r = 0; r < MAXM; ++r)
                if (r < t) mean[r] += upd[t * m + r] * delta;
        }
    } else {
        double *mean = mean_ws + idx * m;
        for (uint32_t r = 0; r < m; ++r)
            mean[r] = c_scale * static_cast<double>(centered_residue(
                                    tp2[((static_cast<size_t>(r) * cols + col) * L) * N + i], q0));
        for (int t = static_cast<int>(m) - 1; t >= 0; --t) {
            const double mu = mean[t];
            const int64_t z = sample_integer_karney(rng, mu, sqrt_var[t]);
            for (uint32_t l = 0; l < L; ++l)
                out[((static_cast<size_t>(t) * cols + col) * L + l) * N + i] =
                    signed_to_residue_mu<W>(z, limbs[l].q, limbs[l].mu64);
            const double delta = static_cast<double>(z) - mu;
            for (int r = 0; r < t; ++r) mean[r] += upd[t * m + r] * delta;
        }
    }
}

__global__ void p1_divisor_kernel(KarneyDivisor *__restrict__ div, const double *__restrict__ sqrt_var, size_t count) {
    const size_t idx = item_index();
    if (idx < count) div[idx] = karney_divisor(sqrt_var[idx]);
}

// persistent-lane form for m <= 4 (rng.h): element = (column, coefficient), m dependent Karney
// integers each (rows m-1 .. 0); integers go to the int64 staging array [row][col][N].  Keystream refills every
// second checkpoint (fill_every; M3A: 0.47 -> 0.445 ms, every third 0.447)
// SEG: the columns are independently seeded segments (rng.h, RngSegments).  The launcher picks a lane count that keeps
// every wave's chunk inside one column (64 * per_lane divides n), so the segment's sub-key and the column's index inside
// its segment - the stream's first word - are per-wave scalars.
template <typename W, int MAXM, int SV, bool SEG = false>
__global__ void __launch_bounds__(SAMPLER_THREADS) p1_sample_lanes_kernel(int64_t *__restrict__ stage, const W *__restrict__ tp2,
                                       const double *__restrict__ sqrt_var_base, const double *__restrict__ update_base,
                                       const KarneyDivisor *__restrict__ div_base, uint32_t m, uint32_t cols, uint32_t L,
                                       uint32_t logN, uint64_t q0, double c_scale, ChaChaKey key, size_t total,
                                       uint32_t per_lane, uint32_t fill_every, SegArg<SEG> segs) {
    __shared__ uint32_t ring[SAMPLER_THREADS * RNG_RING_SLOTS];  // 128 bytes per lane: 8 KB per one-wave workgroup
    WaveChunk chunk = wave_chunk(total, per_lane);
    uint32_t seg_col0 = 0;  // SEG: first column of the chunk's segment
    if constexpr (SEG) {
        const uint32_t col_u = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(chunk.base >> logN)));
        const uint32_t j = rng_segment_of(segs, col_u < cols ? col_u : cols - 1);
        key = segs.key[j];
        seg_col0 = segs.start[j];
    }
    ChaChaRng rng;
    rng_init_keyed(rng, ring, key, 0, 0, SAMPLER_THREADS);
    KarneyFsm f;
    karney_reset(f);
    bool fin = true, have = false;
    uint32_t t = 0, col = 0, i = 0;
    double mean[MAXM], mu = 0.0;
#pragma unroll
    for (int r = 0; r < MAXM; ++r) mean[r] = 0.0;

    // supersteps of 8 as in gauss_samp_lanes_kernel
    auto integer_ready = [&]() {
        if (f.st == KS_DONE && !fin) {  // the integer of row t is ready
            const int64_t z = f.result;
            stage[((static_cast<size_t>(t) * cols + col) << logN) + i] = z;
            const double delta = static_cast<double>(z) - mu;
            const double *upd = update_base + (static_cast<size_t>(i) * m + t) * m;
#pragma unroll
            for (int r = 0; r < MAXM; ++r)
                if (r < (int)t) mean[r] += upd[r] * delta;
            if (t == 0) {
                fin = true;
            } else {
                --t;
#pragma unroll
                for (int r = 0; r < MAXM; ++r)
                    if (r == (int)t) mu = mean[r];
                const size_t sv = static_cast<size_t>(i) * m + t;
                karney_begin(f, mu, sqrt_var_base[sv], div_base[sv]);
            }
        }
    };
    auto element_done = [&]() {
        const bool take = f.st == KS_DONE && fin;
        const uint32_t e = wave_take(chunk, take);
        if (take) {
            have = e < chunk.len;
            if (have) {
                const size_t idx = chunk.base + e;
                col = static_cast<uint32_t>(idx >> logN);
                i = static_cast<uint32_t>(idx & ((1u << logN) - 1));
#pragma unroll
                for (int r = 0; r < MAXM; ++r)
                    if (r < (int)m)
                        mean[r] = c_scale * static_cast<double>(centered_residue(
                                                tp2[(((static_cast<size_t>(r) * cols + col) * L) << logN) + i], q0));
                rng_reopen(rng, static_cast<uint64_t>(col - seg_col0) + 1, static_cast<uint64_t>(i) + 1);
                t = m - 1;
#pragma unroll
                for (int r = 0; r < MAXM; ++r)
                    if (r == (int)t) mu = mean[r];
                fin = false;
                const size_t sv = static_cast<size_t>(i) * m + t;
                karney_begin(f, mu, sqrt_var_base[sv], div_base[sv]);
            } else {
                f.st = KS_IDLE;
            }
        }
    };
    for (uint32_t step = 0;; ++step) {  // one superstep per iteration
        integer_ready();
        element_done();
        if (__all(f.st == KS_IDLE)) break;
        rng_fill_wave(rng, f.st != KS_IDLE, step % fill_every == 0, 16, karney_urgent(SV));
        karney_heavy(f, rng);
#pragma unroll
        for (int s4 = 0; s4 < KARNEY_SUPERSTEP / SV; ++s4) karney_light(f, rng);
#pragma unroll
        for (int sv = 1; sv < SV; ++sv) {
            integer_ready();
            karney_heavy(f, rng);
#pragma unroll
            for (int s4 = 0; s4 < KARNEY_SUPERSTEP / SV; ++s4) karney_light(f, rng);
        }
    }
}

static int check_p1_inputs(const GpuMatrix *a, const GpuMatrix *b, const GpuMatrix *d, double sigma, double s,
                           double dgg_stddev, const char *who) {
    if (!a || !b || !d) return set_error(std::string("invalid ") + who + " arguments");
    if (!(sigma > 0.0) || !(s > sigma)) return set_error(std::string("invalid sigma/s in ") + who);
    if (!(dgg_stddev > 0.0)) return set_error(std::string("dgg_stddev must be positive in ") + who);
    if (a->ctx != b->ctx || a->ctx != d->ctx) return set_error(std::string("context mismatch in ") + who);
    if (a->level != b->level || a->level != d->level) return set_error(std::string("level mismatch in ") + who);
    const size_t dr = a->rows;
    if (a->cols != dr || b->rows != dr || b->cols != dr || d->rows != dr || d->cols != dr)
        return set_error(std::string("A/B/D must be dxd in ") + who);
    if (a->format != GPU_POLY_FORMAT_COEFF || b->format != GPU_POLY_FORMAT_COEFF || d->format != GPU_POLY_FORMAT_COEFF)
        return set_error(std::string("A/B/D must be in Coeff format in ") + who);
    return 0;
}

extern "C" int gpu_matrix_create_p1_covariance_cache(const GpuMatrix *a_mat, const GpuMatrix *b_mat,
                                                     const GpuMatrix *d_mat, double sigma, double s, double dgg_stddev,
                                                     GpuP1CovarianceCache **out_cache) {
    ABI_GUARD_BEGIN
    if (!out_cache) return set_error("gpu_matrix_create_p1_covariance_cache: null out_cache");
    *out_cache = nullptr;
    if (check_p1_inputs(a_mat, b_mat, d_mat, sigma, s, dgg_stddev, "gpu_matrix_create_p1_covariance_cache")) return 1;
    GpuContext *ctx = a_mat->ctx;
    const size_t d = a_mat->rows, m = 2 * d, n = static_cast<size_t>(ctx->N);
    GpuP1CovarianceCache *cache = new GpuP1CovarianceCache();
    cache->ctx = ctx;
    cache->level = a_mat->level;
    cache->d = d;
    cache->m = m;
    cache->n = n;
    cache->sigma = sigma;
    cache->s = s;
    cache->dgg_stddev = dgg_stddev;
    if (d == 0) {
        *out_cache = cache;
        return 0;
    }
    if (ctx_activate(ctx)) {
        delete cache;
        return 1;
    }
    void *cov_ws = nullptr, *sv = nullptr, *uc = nullptr, *kd = nullptr;
    if (ctx_alloc(ctx, n * m * m * sizeof(double), &cov_ws) || ctx_alloc(ctx, n * m * sizeof(double), &sv) ||
        ctx_alloc(ctx, n * m * m * sizeof(double), &uc) || ctx_alloc(ctx, n * m * sizeof(KarneyDivisor), &kd)) {
        ctx_free(ctx, cov_ws);
        ctx_free(ctx, sv);
        ctx_free(ctx, uc);
        delete cache;
        return 1;
    }
    cache->sqrt_var = static_cast<double *>(sv);
    cache->update_coeff = static_cast<double *>(uc);
    cache->karney_div = kd;
    {
        const hipError_t me = hipMemsetAsync(uc, 0, n * m * m * sizeof(double), ctx->stream);
        if (me != hipSuccess) {
            ctx_free(ctx, cov_ws);
            gpu_matrix_destroy_p1_covariance_cache(cache);
            return set_error(me, "hipMemsetAsync");
        }
    }
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(a_mat));
    const unsigned blocks = static_cast<unsigned>((n + 127) / 128);
    if (ctx->wide)
        MXX_LAUNCH(p1_covariance_kernel<uint64_t>, dim3(blocks), dim3(128), 0, ctx->stream,
                           static_cast<const uint64_t *>(a_mat->data), static_cast<const uint64_t *>(b_mat->data),
                           static_cast<const uint64_t *>(d_mat->data), (uint32_t)d, L, (uint32_t)n, ctx->moduli[0],
                           sigma, s, dgg_stddev, static_cast<double *>(cov_ws), cache->sqrt_var, cache->update_coeff);
    else
        MXX_LAUNCH(p1_covariance_kernel<uint32_t>, dim3(blocks), dim3(128), 0, ctx->stream,
                           static_cast<const uint32_t *>(a_mat->data), static_cast<const uint32_t *>(b_mat->data),
                           static_cast<const uint32_t *>(d_mat->data), (uint32_t)d, L, (uint32_t)n, ctx->moduli[0],
                           sigma, s, dgg_stddev, static_cast<double *>(cov_ws), cache->sqrt_var, cache->update_coeff);
    MXX_LAUNCH(p1_divisor_kernel, dim3(static_cast<unsigned>((n * m + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<KarneyDivisor *>(kd), cache->sqrt_var, n * m);
    hipError_t e = hipGetLastError();
    ctx_free(ctx, cov_ws);
    if (e != hipSuccess) {
        gpu_matrix_destroy_p1_covariance_cache(cache);
        return set_error(e, "p1_covariance_kernel");
    }
    *out_cache = cache;
    return 0;
    ABI_GUARD_END
}

extern "C" void gpu_matrix_destroy_p1_covariance_cache(GpuP1CovarianceCache *cache) {
    if (!cache) return;
    if (cache->ctx) {
        (void)hipSetDevice(cache->ctx->device);
        ctx_free(cache->ctx, cache->sqrt_var);
        ctx_free(cache->ctx, cache->update_coeff);
        ctx_free(cache->ctx, cache->karney_div);
    }
    delete cache;
}

static int sample_p1_impl(const GpuP1CovarianceCache *cache, const GpuMatrix *tp2, GpuRngSeed seed, GpuMatrix *out, const RngSegments *segs);

extern "C" int gpu_matrix_sample_p1_full_cached(const GpuP1CovarianceCache *cache, const GpuMatrix *tp2,
                                                GpuRngSeed seed, GpuMatrix *out) {
    ABI_GUARD_BEGIN
    return sample_p1_impl(cache, tp2, seed, out, nullptr);
    ABI_GUARD_END
}

// The same over independently seeded column segments (rng.h, RngSegments): columns [sum seg_cols[<j], + seg_cols[j]) of
// `out` equal what the plain entry point writes for those columns of tp2 taken as a matrix of their own under seeds[j].
extern "C" int gpupoly_matrix_sample_p1_full_cached_segments(const GpuP1CovarianceCache *cache, const GpuMatrix *tp2,
                                                             const GpuRngSeed *seeds, const size_t *seg_cols, size_t nseg,
                                                             GpuMatrix *out) {
    ABI_GUARD_BEGIN
    if (!cache || !tp2 || !out) return set_error("invalid gpupoly_matrix_sample_p1_full_cached_segments arguments");
    if (cache->m > 4 || cache->ctx->env.p1_simple)
        return set_error("gpupoly_matrix_sample_p1_full_cached_segments: unsupported: trapdoor dimension above 2 (no lane kernel)");
    if (static_cast<size_t>(cache->ctx->N) % SAMPLER_THREADS)
        return set_error("gpupoly_matrix_sample_p1_full_cached_segments: unsupported: ring too small for a wave per column chunk");
    RngSegments segs;
    if (!rng_segments_build(segs, seeds, seg_cols, nseg, kTagP1, tp2->cols))
        return set_error("gpupoly_matrix_sample_p1_full_cached_segments: segments must be 1..64, non-empty and cover tp2's columns");
    return sample_p1_impl(cache, tp2, GpuRngSeed{}, out, &segs);
    ABI_GUARD_END
}

static int sample_p1_impl(const GpuP1CovarianceCache *cache, const GpuMatrix *tp2, GpuRngSeed seed, GpuMatrix *out,
                          const RngSegments *segs) {
    {
    if (!cache || !tp2 || !out) return set_error("invalid gpu_matrix_sample_p1_full_cached arguments");
    GpuContext *ctx = cache->ctx;
    if (tp2->ctx != ctx || out->ctx != ctx) return set_error("context mismatch in gpu_matrix_sample_p1_full_cached");
    if (tp2->level != cache->level || out->level != cache->level)
        return set_error("level mismatch in gpu_matrix_sample_p1_full_cached");
    const size_t m = cache->m, cols = tp2->cols;
    if (tp2->rows != m || out->rows != m || out->cols != cols)
        return set_error("tp2/out shape mismatch in gpu_matrix_sample_p1_full_cached");
    if (cols == 0 || m == 0) {
        out->format = GPU_POLY_FORMAT_EVAL;
        return 0;
    }
    if (tp2->format != GPU_POLY_FORMAT_COEFF)
        return set_error("tp2 must be in Coeff format in gpu_matrix_sample_p1_full_cached");
    const double denom = cache->s * cache->s - cache->sigma * cache->sigma;
    if (!(denom > 0.0)) return set_error("invalid cached Gaussian denominator");
    const double c_scale = -(cache->sigma * cache->sigma) / denom;
    if (ctx_activate(ctx)) return 1;
    const uint32_t L = static_cast<uint32_t>(matrix_limbs(out)), N = static_cast<uint32_t>(ctx->N);
    const size_t total = cols * static_cast<size_t>(N);
    // MXX_HIP_P1=simple keeps the one-thread-per-element kernel for every m (A/B runs, tests)
    if (m <= 4 && !ctx->env.p1_simple) {
        void *stage = nullptr;
        if (ctx_alloc(ctx, total * m * sizeof(int64_t), &stage)) return 1;
        const ChaChaKey key = chacha_subkey(seed, 0, kTagP1);
        // per (column, coefficient): m limb-0 residues of tp2, m standard deviations + divisors and m^2 update coefficients of
        // the cache (the cache is per coefficient: re-read per column from L2 / the Infinity Cache), m int64 samples written
        MXX_TRACE_BYTES(static_cast<double>(total) * m * (ctx->word_bytes + 8.0) +
                        static_cast<double>(N) * m * (8.0 + sizeof(KarneyDivisor) + 8.0 * m));
#define LAUNCH_P1K(WT, MAXM, SV, SEG, SEGARG)                                                                     \
    MXX_LAUNCH((p1_sample_lanes_kernel<WT, MAXM, SV, SEG>), dim3(lblocks), dim3(SAMPLER_THREADS), 0, ctx->stream, \
               static_cast<int64_t *>(stage), static_cast<const WT *>(tp2->data), cache->sqrt_var,                \
               cache->update_coeff, static_cast<const KarneyDivisor *>(cache->karney_div), (uint32_t)m,           \
               (uint32_t)cols, L, ctx->logN, ctx->moduli[0], c_scale, key, total, per_lane,                       \
               static_cast<uint32_t>(ctx->env.sampler_fill_every ? ctx->env.sampler_fill_every : 2), SEGARG)
#define LAUNCH_P1L(WT, MAXM)                                                                                      \
    do {                                                                                                          \
        uint32_t per_lane =                                                                                       \
            sampler_per_lane(total, reinterpret_cast<const void *>(p1_sample_lanes_kernel<WT, MAXM, KARNEY_SERVICES>), ctx->device, ctx->env.sampler_per_lane); \
        if (segs) /* a divisor of n / 64: every wave's chunk inside one column, i.e. inside one segment */        \
            while ((static_cast<size_t>(N) / SAMPLER_THREADS) % per_lane) --per_lane;                             \
        const unsigned lblocks = static_cast<unsigned>((total + SAMPLER_THREADS * per_lane - 1) / (SAMPLER_THREADS * per_lane)); \
        if (segs) {                                                                                               \
            if (per_lane == 1) LAUNCH_P1K(WT, MAXM, 1, true, *segs);                                              \
            else LAUNCH_P1K(WT, MAXM, KARNEY_SERVICES, true, *segs);                                              \
        } else if (per_lane == 1) {                                                                               \
            LAUNCH_P1K(WT, MAXM, 1, false, NoSegments{});                                                         \
        } else {                                                                                                  \
            LAUNCH_P1K(WT, MAXM, KARNEY_SERVICES, false, NoSegments{});                                           \
        }                                                                                                         \
    } while (0)
        if (ctx->wide) {
            if (m <= 2) LAUNCH_P1L(uint64_t, 2);
            else LAUNCH_P1L(uint64_t, 4);
        } else {
            if (m <= 2) LAUNCH_P1L(uint32_t, 2);
            else LAUNCH_P1L(uint32_t, 4);
        }
#undef LAUNCH_P1L
#undef LAUNCH_P1K
        const hipError_t le = hipGetLastError();
        int lrc = le == hipSuccess ? launch_scatter_i64(out, static_cast<const int64_t *>(stage)) : 0;
        ctx_free(ctx, stage);
        if (le != hipSuccess) return set_error(le, "p1_sample_lanes_kernel");
        if (lrc) return lrc;
        lrc = launch_ntt(ctx, out->data, matrix_polys(out) * L, static_cast<int>(L), false);
        if (lrc) return lrc;
        out->format = GPU_POLY_FORMAT_EVAL;
        return 0;
    }
    const dim3 blocks = item_grid(total, 128);
    void *mean_ws = nullptr;
    if (m > 8 && ctx_alloc(ctx, total * m * sizeof(double), &mean_ws)) return 1;
#define LAUNCH_P1(WT, MAXM)                                                                                       \
    MXX_LAUNCH((p1_sample_kernel<WT, MAXM>), dim3(blocks), dim3(128), 0, ctx->stream,                      \
                       static_cast<WT *>(out->data), static_cast<const WT *>(tp2->data), ctx->d_limbs,             \
                       cache->sqrt_var, cache->update_coeff, (uint32_t)m, (uint32_t)cols, L, N, ctx->moduli[0],    \
                       c_scale, seed, static_cast<double *>(mean_ws))
    if (ctx->wide) {
        if (m <= 2) LAUNCH_P1(uint64_t, 2);
        else if (m <= 4) LAUNCH_P1(uint64_t, 4);
        else if (m <= 8) LAUNCH_P1(uint64_t, 8);
        else LAUNCH_P1(uint64_t, 0);
    } else {
        if (m <= 2) LAUNCH_P1(uint32_t, 2);
        else if (m <= 4) LAUNCH_P1(uint32_t, 4);
        else if (m <= 8) LAUNCH_P1(uint32_t, 8);
        else LAUNCH_P1(uint32_t, 0);
    }
#undef LAUNCH_P1
    hipError_t e = hipGetLastError();
    if (mean_ws) ctx_free(ctx, mean_ws);
    if (e != hipSuccess) return set_error(e, "p1_sample_kernel");
    // always finishes in EVAL
    int rc = launch_ntt(ctx, out->data, matrix_polys(out) * L, static_cast<int>(L), false);
    if (rc) return rc;
    out->format = GPU_POLY_FORMAT_EVAL;
    return 0;
    }
}

extern "C" int gpu_matrix_sample_p1_full(const GpuMatrix *a_mat, const GpuMatrix *b_mat, const GpuMatrix *d_mat,
                                         const GpuMatrix *tp2, double sigma, double s, double dgg_stddev,
                                         GpuRngSeed seed, GpuMatrix *out) {
    ABI_GUARD_BEGIN
    GpuP1CovarianceCache *cache = nullptr;
    int rc = gpu_matrix_create_p1_covariance_cache(a_mat, b_mat, d_mat, sigma, s, dgg_stddev, &cache);
    if (rc) return rc;
    rc = gpu_matrix_sample_p1_full_cached(cache, tp2, seed, out);
    gpu_matrix_destroy_p1_covariance_cache(cache);
    return rc;
    ABI_GUARD_END
}
