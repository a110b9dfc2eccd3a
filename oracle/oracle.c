/*
 * oracle.c — CPU restatement of the mxx DCRT ring-matrix hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (mxx_amd/, the C-ABI
 * library) may link, import or call this file.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() use it, and only as the
 * checker / the reported CPU baseline.
 *
 * PARITY UNPINNED at the byte level: the reference's CPU arithmetic lives in
 * OpenFHE (MachinaIO fork, reached through crate `openfhe` 0.3.2,
 * git MachinaIO/openfhe-rs@9c9d81ce, Cargo.lock:514-523), which is not
 * vendored and cannot be built here, and the reference holds no golden vectors
 * for this path (all its hot-path tests are algebraic).  This
 * file therefore restates OpenFHE's published algorithm following the
 * reference's own in-tree restatement and call sites, and is pinned by
 *   (1) the reference's algebraic predicates (G·G^-1(M)=M, A·x=u, round trips),
 *   (2) a convention-independent schoolbook negacyclic product, and
 *   (3) two independent implementations (plain `%` vs Shoup) agreeing.
 *
 * Conventions followed (file:line relative to /root/reference):
 *   - psi = MIN over all primitive 2n-th roots mod q
 *       src/gadgets/ntt/mod.rs:96-128
 *   - tables fwd[bitrev(i)] = psi^i, inv[bitrev(i)] = psi^-i
 *       src/gadgets/ntt/mod.rs:189-198
 *   - forward = Cooley-Tukey, natural in -> bit-reversed out
 *       src/gadgets/ntt/mod.rs:288-339
 *   - inverse = Gentleman-Sande + n^-1, bit-reversed in -> natural out
 *       src/gadgets/ntt/mod.rs:341-392
 *   - matrix layout on the wire: [poly][limb][n] u64, poly = row*cols+col
 *       src/poly/dcrt/gpu.rs:758-788, src/matrix/gpu_dcrt_poly.rs:781-816
 *   - matrix product semantics   src/matrix/base/memory.rs:450-480,589-605
 *   - digit decomposition + last-digit mask
 *       src/matrix/dcrt_poly.rs:134-198,453-493, src/poly/dcrt/params.rs:77-98
 *       cuda/src/matrix/MatrixDecompose.cu:77-113
 *   - gadget entries             cuda/src/matrix/MatrixDecompose.cu:235-252
 *   - CRT basis rule (OpenFHE ILDCRTParams(order=2n, depth, bits):
 *       LastPrime then PreviousPrime; src/poly/dcrt/params.rs:60-66)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ */
/* scalar number theory                                                */
/* ------------------------------------------------------------------ */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

uint64_t orc_powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    while (e) {
        if (e & 1) r = orc_mulmod(r, b, q);
        b = orc_mulmod(b, b, q);
        e >>= 1;
    }
    return r;
}

/* q prime */
uint64_t orc_invmod(uint64_t a, uint64_t q) { return orc_powmod(a % q, q - 2, q); }

/* deterministic Miller-Rabin for 64-bit */
int orc_is_prime(uint64_t n) {
    static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return 0;
    for (size_t i = 0; i < 12; ++i) {
        if (n % bases[i] == 0) return n == bases[i];
    }
    uint64_t d = n - 1;
    int s = 0;
    while ((d & 1) == 0) { d >>= 1; ++s; }
    for (size_t i = 0; i < 12; ++i) {
        uint64_t x = orc_powmod(bases[i], d, n);
        if (x == 1 || x == n - 1) continue;
        int comp = 1;
        for (int r = 1; r < s; ++r) {
            x = orc_mulmod(x, x, n);
            if (x == n - 1) { comp = 0; break; }
        }
        if (comp) return 0;
    }
    return 1;
}

/* OpenFHE ILDCRTParams(order = 2n, depth, bits): q0 = LastPrime(bits, 2n) =
 * largest prime < 2^bits with q = 1 (mod 2n); q_{i+1} = PreviousPrime(q_i, 2n).
 * Returns 0 on success. */
int orc_gen_crt_basis(uint32_t n, uint32_t depth, uint32_t bits, uint64_t *out) {
    if (bits < 2 || bits > 62) return -1;
    uint64_t m = 2ull * n;
    uint64_t q = (1ull << bits) + 1; /* 2^bits is a multiple of 2n */
    for (uint32_t i = 0; i < depth; ++i) {
        do {
            if (q <= m) return -2;
            q -= m;
        } while (!orc_is_prime(q));
        if ((q >> (bits - 1)) != 1) return -3; /* fell below `bits` bits */
        out[i] = q;
    }
    return 0;
}

/* psi = min over all primitive `order`-th roots (order a power of two dividing q-1) */
uint64_t orc_min_primitive_root(uint64_t q, uint64_t order) {
    if (order == 1) return 1;
    uint64_t qm1 = q - 1;
    int v = __builtin_ctzll(qm1);
    int want = __builtin_ctzll(order);
    if (want > v) return 0;
    uint64_t odd = qm1 >> v;
    uint64_t maximal = 0;
    for (uint64_t x = 2; x < q; ++x) {
        uint64_t r = orc_powmod(x, odd, q); /* order divides 2^v */
        if (orc_powmod(r, 1ull << (v - 1), q) != 1) { maximal = r; break; }
    }
    if (!maximal) return 0;
    uint64_t root = orc_powmod(maximal, 1ull << (v - want), q); /* primitive order-th root */
    uint64_t sq = orc_mulmod(root, root, q);
    uint64_t cur = root, best = root;
    for (uint64_t i = 1; i < order / 2; ++i) {
        cur = orc_mulmod(cur, sq, q); /* root^(2i+1) */
        if (cur < best) best = cur;
    }
    return best;
}

static inline uint32_t bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

static inline uint32_t ilog2(uint32_t n) { return 31 - __builtin_clz(n); }

/* fwd[bitrev(i)] = psi^i ; inv[bitrev(i)] = psi^-i ; *n_inv = n^-1 */
void orc_ntt_tables(uint64_t q, uint32_t n, uint64_t *fwd, uint64_t *inv, uint64_t *n_inv) {
    uint64_t psi = orc_min_primitive_root(q, 2ull * n);
    uint64_t ipsi = orc_invmod(psi, q);
    uint32_t bits = ilog2(n);
    uint64_t p = 1, ip = 1;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t r = bitrev(i, bits);
        fwd[r] = p;
        inv[r] = ip;
        p = orc_mulmod(p, psi, q);
        ip = orc_mulmod(ip, ipsi, q);
    }
    *n_inv = orc_invmod(n % q, q);
}

/* plain (`%`-based) transforms: the readable restatement */
void orc_ntt_forward_plain(uint64_t *x, uint32_t n, uint64_t q, const uint64_t *fwd) {
    uint32_t t = n >> 1;
    for (uint32_t m = 1; m < n; m <<= 1, t >>= 1) {
        for (uint32_t i = 0; i < m; ++i) {
            uint64_t w = fwd[m + i];
            uint32_t j1 = 2 * i * t;
            for (uint32_t j = j1; j < j1 + t; ++j) {
                uint64_t u = x[j];
                uint64_t v = orc_mulmod(x[j + t], w, q);
                uint64_t s = u + v; if (s >= q) s -= q;
                uint64_t d = u >= v ? u - v : u + q - v;
                x[j] = s;
                x[j + t] = d;
            }
        }
    }
}

void orc_ntt_inverse_plain(uint64_t *x, uint32_t n, uint64_t q, const uint64_t *inv, uint64_t n_inv) {
    uint32_t t = 1;
    for (uint32_t m = n >> 1; m >= 1; m >>= 1, t <<= 1) {
        for (uint32_t i = 0; i < m; ++i) {
            uint64_t w = inv[m + i];
            uint32_t j1 = 2 * i * t;
            for (uint32_t j = j1; j < j1 + t; ++j) {
                uint64_t u = x[j];
                uint64_t v = x[j + t];
                uint64_t s = u + v; s -= q & (uint64_t)-(int64_t)(s >= q);
                uint64_t d = u - v; d += q & (uint64_t)-(int64_t)(u < v);
                x[j] = s;
                x[j + t] = orc_mulmod(d, w, q);
            }
        }
    }
    for (uint32_t j = 0; j < n; ++j) x[j] = orc_mulmod(x[j], n_inv, q);
}

/* Shoup companions: w' = floor(w * 2^64 / q).  Fast path used as CPU baseline
 * (OpenFHE itself uses precomputed-constant modular multiplies in its NTT). */
static inline uint64_t shoup(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }
static inline uint64_t mul_shoup(uint64_t x, uint64_t w, uint64_t ws, uint64_t q) {
    uint64_t t = (uint64_t)(((u128)x * ws) >> 64);
    uint64_t r = x * w - t * q;
    return r - (q & (uint64_t)-(int64_t)(r >= q));
}

static void ntt_forward_shoup(uint64_t *x, uint32_t n, uint64_t q, const uint64_t *fwd, const uint64_t *fwds) {
    uint32_t t = n >> 1;
    for (uint32_t m = 1; m < n; m <<= 1, t >>= 1) {
        for (uint32_t i = 0; i < m; ++i) {
            uint64_t w = fwd[m + i], ws = fwds[m + i];
            uint32_t j1 = 2 * i * t;
            for (uint32_t j = j1; j < j1 + t; ++j) {
                uint64_t u = x[j];
                uint64_t v = mul_shoup(x[j + t], w, ws, q);
                /* mask form: both corrections are coin flips on uniform residues, and gcc turned the ?: forms of this loop into
                 * branches (3x the inverse transform's time, all of it mispredictions) */
                uint64_t s = u + v; s -= q & (uint64_t)-(int64_t)(s >= q);
                uint64_t d = u - v; d += q & (uint64_t)-(int64_t)(u < v);
                x[j] = s;
                x[j + t] = d;
            }
        }
    }
}

static void ntt_inverse_shoup(uint64_t *x, uint32_t n, uint64_t q, const uint64_t *inv, const uint64_t *invs,
                              uint64_t n_inv) {
    uint32_t t = 1;
    for (uint32_t m = n >> 1; m >= 1; m >>= 1, t <<= 1) {
        for (uint32_t i = 0; i < m; ++i) {
            uint64_t w = inv[m + i], ws = invs[m + i];
            uint32_t j1 = 2 * i * t;
            for (uint32_t j = j1; j < j1 + t; ++j) {
                uint64_t u = x[j];
                uint64_t v = x[j + t];
                uint64_t s = u + v; s -= q & (uint64_t)-(int64_t)(s >= q);
                uint64_t d = u - v; d += q & (uint64_t)-(int64_t)(u < v);
                x[j] = s;
                x[j + t] = mul_shoup(d, w, ws, q);
            }
        }
    }
    uint64_t ns = shoup(n_inv, q);
    for (uint32_t j = 0; j < n; ++j) x[j] = mul_shoup(x[j], n_inv, ns, q);
}

/* ------------------------------------------------------------------ */
/* table cache keyed by (q, n)                                         */
/* ------------------------------------------------------------------ */
typedef struct {
    uint64_t q;
    uint32_t n;
    uint64_t *fwd, *fwds, *inv, *invs;
    uint64_t n_inv;
} tab_t;
/* entries are allocated one by one and never move (callers keep the pointer); the index grows by doubling - a fixed
 * 256-entry array overflowed (NULL table, crash) once the GPU suite visited more than 256 (modulus, ring) pairs */
static tab_t **g_tabs = NULL;
static int g_ntabs = 0, g_tabs_cap = 0;

static const tab_t *get_tab(uint64_t q, uint32_t n) {
    const tab_t *found = NULL;
#pragma omp critical(orc_tab)
    {
        for (int i = 0; i < g_ntabs; ++i)
            if (g_tabs[i]->q == q && g_tabs[i]->n == n) { found = g_tabs[i]; break; }
        if (!found) {
            if (g_ntabs == g_tabs_cap) {
                g_tabs_cap = g_tabs_cap ? 2 * g_tabs_cap : 64;
                g_tabs = realloc(g_tabs, sizeof(tab_t *) * (size_t)g_tabs_cap);
            }
            tab_t *t = malloc(sizeof(tab_t));
            g_tabs[g_ntabs] = t;
            t->q = q; t->n = n;
            t->fwd = malloc(sizeof(uint64_t) * n); t->fwds = malloc(sizeof(uint64_t) * n);
            t->inv = malloc(sizeof(uint64_t) * n); t->invs = malloc(sizeof(uint64_t) * n);
            orc_ntt_tables(q, n, t->fwd, t->inv, &t->n_inv);
            for (uint32_t i = 0; i < n; ++i) { t->fwds[i] = shoup(t->fwd[i], q); t->invs[i] = shoup(t->inv[i], q); }
            ++g_ntabs;
            found = t;
        }
    }
    return found;
}

/* single-vector transforms through the cache; plain!=0 selects the `%` variant */
void orc_ntt(uint64_t *x, uint32_t n, uint64_t q, int inverse, int plain) {
    const tab_t *t = get_tab(q, n);
    if (plain) {
        if (inverse) orc_ntt_inverse_plain(x, n, q, t->inv, t->n_inv);
        else orc_ntt_forward_plain(x, n, q, t->fwd);
    } else {
        if (inverse) ntt_inverse_shoup(x, n, q, t->inv, t->invs, t->n_inv);
        else ntt_forward_shoup(x, n, q, t->fwd, t->fwds);
    }
}

void orc_set_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------ */
/* matrix-level ops on the wire layout [poly][limb][n] u64              */
/* ------------------------------------------------------------------ */
void orc_matrix_ntt(uint64_t *data, size_t polys, uint32_t L, uint32_t n, const uint64_t *moduli, int inverse) {
    for (uint32_t l = 0; l < L; ++l) (void)get_tab(moduli[l], n);
    long total = (long)(polys * L);
#pragma omp parallel for schedule(static)
    for (long v = 0; v < total; ++v) {
        uint32_t l = (uint32_t)(v % L);
        orc_ntt(data + (size_t)v * n, n, moduli[l], inverse, 0);
    }
}

/* op: 0 add, 1 sub, 2 mul.  b_polys == 1 broadcasts b (scalar poly). */
void orc_pointwise(int op, uint64_t *out, const uint64_t *a, const uint64_t *b, size_t polys, size_t b_polys,
                   uint32_t L, uint32_t n, const uint64_t *moduli) {
    long total = (long)(polys * L);
#pragma omp parallel for schedule(static)
    for (long v = 0; v < total; ++v) {
        uint32_t l = (uint32_t)(v % L);
        size_t p = (size_t)v / L;
        uint64_t q = moduli[l];
        const uint64_t *av = a + (size_t)v * n;
        const uint64_t *bv = b + ((b_polys == 1 ? 0 : p) * L + l) * (size_t)n;
        uint64_t *ov = out + (size_t)v * n;
        if (op == 2 && !(q >> 32)) {
            /* word-sized modulus: Barrett with mu = floor(2^64 / q) instead of a 128-bit division */
            const uint64_t mu = (uint64_t)((((u128)1) << 64) / q);
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t prod = av[i] * bv[i];
                uint64_t r = prod - (uint64_t)(((u128)prod * mu) >> 64) * q;
                ov[i] = r >= q ? r - q : r;
            }
            continue;
        }
        for (uint32_t i = 0; i < n; ++i) {
            uint64_t x = av[i], y = bv[i], r;
            if (op == 0) { r = x + y; if (r >= q) r -= q; }
            else if (op == 1) { r = x >= y ? x - y : x + q - y; }
            else r = orc_mulmod(x, y, q);
            ov[i] = r;
        }
    }
}

/* C(rows x cols) = A(rows x inner) * B(inner x cols), every entry an EVAL-format
 * DCRT poly: per limb, per slot an integer mat-mul mod q. */
void orc_matmul(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t rows, size_t inner, size_t cols,
                uint32_t L, uint32_t n, const uint64_t *moduli) {
    long total = (long)(rows * cols * L);
#pragma omp parallel for schedule(static)
    for (long v = 0; v < total; ++v) {
        uint32_t l = (uint32_t)(v % L);
        size_t rc = (size_t)v / L;
        size_t r = rc / cols, c = rc % cols;
        uint64_t q = moduli[l];
        uint64_t *ov = out + (size_t)v * n;
        /* lazy accumulation: terms < q^2; flush before 128-bit overflow is impossible for q < 2^62
         * and inner < 2^4; keep it simple and reduce each term into a u128 sum of reduced products */
        for (uint32_t i = 0; i < n; ++i) {
            u128 acc = 0;
            for (size_t k = 0; k < inner; ++k) {
                uint64_t x = a[((r * inner + k) * L + l) * (size_t)n + i];
                uint64_t y = b[((k * cols + c) * L + l) * (size_t)n + i];
                acc += (u128)x * y;
                if ((k & 7) == 7) acc %= q;
            }
            ov[i] = (uint64_t)(acc % q);
        }
    }
}

/* Same product, cache-friendlier loop order (k outer, slots inner): the CPU-baseline variant.
 * Word-sized moduli (q < 2^32): products < 2^64 are summed lazily in 64-bit accumulators (as many
 * as fit, one Barrett reduction per window; the multiply is the 32x32->64 form compilers vectorise);
 * wider moduli: 128-bit accumulators. */
static void matmul_lazy_u64(uint64_t *ov, const uint64_t *a, const uint64_t *b, size_t r, size_t c, size_t inner,
                            size_t cols, uint32_t L, uint32_t l, uint32_t n, uint64_t q, uint64_t *acc) {
    const u128 q2 = (u128)(q - 1) * (q - 1);
    u128 win = ((((u128)1) << 64) - q) / q2;  /* (q-1) + win*(q-1)^2 < 2^64 */
    const size_t window = win > (1u << 20) ? (1u << 20) : (size_t)win;
    const uint64_t mu = (uint64_t)((((u128)1) << 64) / q);
    for (uint32_t i = 0; i < n; ++i) acc[i] = 0;
    size_t since = 0;
    for (size_t k = 0; k < inner; ++k) {
        const uint64_t *x = a + ((r * inner + k) * L + l) * (size_t)n;
        const uint64_t *y = b + ((k * cols + c) * L + l) * (size_t)n;
        for (uint32_t i = 0; i < n; ++i) acc[i] += (uint64_t)(uint32_t)x[i] * (uint32_t)y[i];
        if (++since == window || k + 1 == inner) {
            for (uint32_t i = 0; i < n; ++i) {
                uint64_t v = acc[i];
                uint64_t red = v - (uint64_t)(((u128)v * mu) >> 64) * q;
                acc[i] = red >= q ? red - q : red;
            }
            since = 0;
        }
    }
    for (uint32_t i = 0; i < n; ++i) ov[i] = acc[i];
}

void orc_matmul_fast(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t rows, size_t inner, size_t cols,
                     uint32_t L, uint32_t n, const uint64_t *moduli) {
    long total = (long)(rows * cols * L);
#pragma omp parallel
    {
        u128 *acc = malloc(sizeof(u128) * n);
#pragma omp for schedule(static)
        for (long v = 0; v < total; ++v) {
            uint32_t l = (uint32_t)(v % L);
            size_t rc = (size_t)v / L;
            size_t r = rc / cols, c = rc % cols;
            uint64_t q = moduli[l];
            if (!(q >> 32) && inner > 0) {
                matmul_lazy_u64(out + (size_t)v * n, a, b, r, c, inner, cols, L, l, n, q, (uint64_t *)acc);
                continue;
            }
            /* how many q^2-bounded terms fit in 128 bits */
            for (uint32_t i = 0; i < n; ++i) acc[i] = 0;
            size_t since = 0;
            for (size_t k = 0; k < inner; ++k) {
                const uint64_t *x = a + ((r * inner + k) * L + l) * (size_t)n;
                const uint64_t *y = b + ((k * cols + c) * L + l) * (size_t)n;
                for (uint32_t i = 0; i < n; ++i) acc[i] += (u128)x[i] * y[i];
                if (++since == 8) {
                    for (uint32_t i = 0; i < n; ++i) acc[i] %= q;
                    since = 0;
                }
            }
            uint64_t *ov = out + (size_t)v * n;
            for (uint32_t i = 0; i < n; ++i) ov[i] = (uint64_t)(acc[i] % q);
        }
        free(acc);
    }
}

/* schoolbook negacyclic product in Z_q[x]/(x^n+1): convention-independent check */
void orc_negacyclic_schoolbook(uint64_t *out, const uint64_t *a, const uint64_t *b, uint32_t n, uint64_t q) {
    for (uint32_t i = 0; i < n; ++i) out[i] = 0;
    for (uint32_t i = 0; i < n; ++i) {
        for (uint32_t j = 0; j < n; ++j) {
            uint64_t p = orc_mulmod(a[i], b[j], q);
            uint32_t k = i + j;
            if (k < n) { out[k] += p; if (out[k] >= q) out[k] -= q; }
            else { k -= n; out[k] = out[k] >= p ? out[k] - p : out[k] + q - p; }
        }
    }
}

static inline uint32_t bit_width(uint64_t v) { return v ? 64 - (uint32_t)__builtin_clzll(v) : 0; }

uint32_t orc_crt_bits(const uint64_t *moduli, uint32_t L) {
    uint32_t b = 0;
    for (uint32_t l = 0; l < L; ++l) { uint32_t w = bit_width(moduli[l]); if (w > b) b = w; }
    return b;
}

/* Digit decomposition, COEFFICIENT domain in and out.
 * src: rows x cols; out: (rows*k) x cols with k = dpt*L (small: k = dpt, limb 0 only).
 * out row r*k + t*dpt + d holds digit d of the limb-t residue, replicated into every limb.
 * The last digit of a tower keeps bits(q_t) - (dpt-1)*base_bits bits
 * (MatrixDecompose.cu:95-103; params.rs:81-89 with crt_bits). */
void orc_decompose(uint64_t *out, const uint64_t *src, size_t rows, size_t cols, uint32_t L, uint32_t n,
                   const uint64_t *moduli, uint32_t base_bits, int small) {
    uint32_t crt_bits = orc_crt_bits(moduli, L);
    uint32_t dpt = (crt_bits + base_bits - 1) / base_bits;
    uint32_t towers = small ? 1 : L;
    size_t k = (size_t)dpt * towers;
    for (size_t r = 0; r < rows; ++r)
        for (size_t c = 0; c < cols; ++c)
            for (uint32_t t = 0; t < towers; ++t) {
                uint32_t src_bits = bit_width(moduli[t]);
                const uint64_t *sv = src + ((r * cols + c) * L + t) * (size_t)n;
                for (uint32_t d = 0; d < dpt; ++d) {
                    uint32_t shift = d * base_bits;
                    uint64_t mask = 0;
                    if (shift < src_bits) {
                        uint32_t rem = src_bits - shift;
                        uint32_t db = base_bits < rem ? base_bits : rem;
                        mask = db >= 64 ? ~0ull : ((1ull << db) - 1);
                    }
                    size_t orow = r * k + (size_t)t * dpt + d;
                    for (uint32_t l = 0; l < L; ++l) {
                        uint64_t *ov = out + ((orow * cols + c) * L + l) * (size_t)n;
                        uint64_t ql = moduli[l];
                        for (uint32_t i = 0; i < n; ++i) {
                            uint64_t digit = shift >= 64 ? 0 : ((sv[i] >> shift) & mask);
                            ov[i] = digit >= ql ? digit % ql : digit;
                        }
                    }
                }
            }
}

/* Gadget matrix G = I_size (x) g, COEFFICIENT domain (constant polys).
 * out: size x size*k.  Regular: entry (t,d) = b^d mod q_t in limb t, 0 in other limbs.
 * Small: k = dpt, b^d in every limb. */
void orc_fill_gadget(uint64_t *out, size_t size, uint32_t L, uint32_t n, const uint64_t *moduli,
                     uint32_t base_bits, int small) {
    uint32_t crt_bits = orc_crt_bits(moduli, L);
    uint32_t dpt = (crt_bits + base_bits - 1) / base_bits;
    size_t k = small ? dpt : (size_t)dpt * L;
    size_t cols = size * k;
    memset(out, 0, sizeof(uint64_t) * size * cols * L * n);
    for (size_t r = 0; r < size; ++r)
        for (size_t j = 0; j < k; ++j) {
            uint32_t tower = (uint32_t)(j / dpt), digit = (uint32_t)(j % dpt);
            size_t col = r * k + j;
            for (uint32_t l = 0; l < L; ++l) {
                if (!small && l != tower) continue;
                out[((r * cols + col) * L + l) * (size_t)n] = orc_powmod(1ull << base_bits, digit, moduli[l]);
            }
        }
}

/* ------------------------------------------------------------------ */
/* CPU-baseline kernels (timed by bench.py's cpu_baseline leg)          */
/* ------------------------------------------------------------------ */
/* one "ring multiplication" step of workload M1: c = INTT(NTT(a) o NTT(b)) over a batch */
void orc_ring_mul_batch(uint64_t *c, uint64_t *a, uint64_t *b, size_t polys, uint32_t L, uint32_t n,
                        const uint64_t *moduli) {
    orc_matrix_ntt(a, polys, L, n, moduli, 0);
    orc_matrix_ntt(b, polys, L, n, moduli, 0);
    orc_pointwise(2, c, a, b, polys, polys, L, n, moduli);
    orc_matrix_ntt(c, polys, L, n, moduli, 1);
}

/* ---- in-C timed baselines: inputs generated here (splitmix64), every repetition timed with
 * CLOCK_MONOTONIC, no host-language work inside the timed region.  Thread count = orc_set_threads. ---- */
#include <time.h>
static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
#include <sys/mman.h>
static uint64_t *bench_random(size_t polys, uint32_t L, uint32_t n, const uint64_t *moduli, uint64_t seed) {
    /* multi-GB operands: 2 MiB-aligned + transparent huge pages keep first-touch faults out of the way */
    void *raw = NULL;
    size_t bytes = sizeof(uint64_t) * polys * L * n;
    if (posix_memalign(&raw, (size_t)2 << 20, bytes ? bytes : 8)) return NULL;
#ifdef MADV_HUGEPAGE
    (void)madvise(raw, bytes, MADV_HUGEPAGE);
#endif
    uint64_t *m = raw;
    long total = (long)(polys * L);
#pragma omp parallel for schedule(static)
    for (long v = 0; v < total; ++v) {
        uint32_t l = (uint32_t)(v % L);
        uint64_t s = seed ^ ((uint64_t)v * 0x100000001b3ull);
        uint64_t *x = m + (size_t)v * n;
        for (uint32_t i = 0; i < n; ++i) x[i] = (uint64_t)(((u128)splitmix64(&s) * moduli[l]) >> 64);  /* uniform in [0, q) */
    }
    return m;
}

/* (rows x inner) * (inner x cols) in EVAL form, inputs generated once; after one warm-up, reps_all
 * repetitions on `threads` threads (sec_all) and reps_one repetitions on one thread (sec_one) */
int orc_bench_matmul(size_t rows, size_t inner, size_t cols, uint32_t L, uint32_t n, const uint64_t *moduli, int threads,
                     int reps_all, double *sec_all, int reps_one, double *sec_one) {
    orc_set_threads(threads);
    uint64_t *a = bench_random(rows * inner, L, n, moduli, 0x6d7878 ^ 4), *b = bench_random(inner * cols, L, n, moduli, 0x6d7878 ^ 5);
    uint64_t *c = malloc(sizeof(uint64_t) * rows * cols * L * n);
    if (!a || !b || !c) { free(a); free(b); free(c); return 1; }
    orc_matmul_fast(c, a, b, rows, inner, cols, L, n, moduli);
    for (int r = 0; r < reps_all; ++r) {
        double t0 = now_s();
        orc_matmul_fast(c, a, b, rows, inner, cols, L, n, moduli);
        sec_all[r] = now_s() - t0;
    }
    orc_set_threads(1);
    for (int r = 0; r < reps_one; ++r) {
        double t0 = now_s();
        orc_matmul_fast(c, a, b, rows, inner, cols, L, n, moduli);
        sec_one[r] = now_s() - t0;
    }
    orc_set_threads(threads);
    free(a); free(b); free(c);
    return 0;
}

/* workload M1 on `polys` polynomials: x <- INTT(NTT(x) o w); sec[3*r + {0,1,2}] = forward NTT,
 * pointwise product, inverse NTT of repetition r; same all-threads / one-thread split */
static void ring_mul_reps(uint64_t *x, const uint64_t *w, size_t polys, uint32_t L, uint32_t n, const uint64_t *moduli,
                          int reps, double *sec) {
    for (int r = -1; r < reps; ++r) {
        double t0 = now_s();
        orc_matrix_ntt(x, polys, L, n, moduli, 0);
        double t1 = now_s();
        orc_pointwise(2, x, x, w, polys, 1, L, n, moduli);
        double t2 = now_s();
        orc_matrix_ntt(x, polys, L, n, moduli, 1);
        double t3 = now_s();
        if (r >= 0) { sec[3 * r] = t1 - t0; sec[3 * r + 1] = t2 - t1; sec[3 * r + 2] = t3 - t2; }
    }
}
int orc_bench_ring_mul(size_t polys, uint32_t L, uint32_t n, const uint64_t *moduli, int threads, int reps_all,
                       double *sec_all, int reps_one, double *sec_one) {
    orc_set_threads(threads);
    uint64_t *x = bench_random(polys, L, n, moduli, 0x6d7878 ^ 2), *w = bench_random(1, L, n, moduli, 0x6d7878 ^ 3);
    if (!x || !w) { free(x); free(w); return 1; }
    ring_mul_reps(x, w, polys, L, n, moduli, reps_all, sec_all);
    orc_set_threads(1);
    ring_mul_reps(x, w, polys, L, n, moduli, reps_one, sec_one);
    orc_set_threads(threads);
    free(x); free(w);
    return 0;
}
