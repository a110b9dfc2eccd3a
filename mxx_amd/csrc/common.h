// common.h — private structures of libgpupoly (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <map>
#include <unordered_map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gpupoly.h"

constexpr size_t GPUPOLY_MAX_LIMBS = 64;  // cuda/include/Runtime.cuh:51 of the reference

// Per-limb constants, one array entry per prime, resident in HBM and read
// through the scalar cache (uniform per workgroup).
struct LimbConst {
    uint64_t q;          // modulus
    uint64_t mu;         // Barrett constant: floor(2^(2*kbits)/q)   (kbits = bits(q))
    uint64_t mu64;       // floor(2^64/q) (u32 path: lazy 64-bit sum reduction)
    uint64_t n_inv;      // N^-1 mod q
    uint64_t n_inv_sh;   // Shoup companion of n_inv (2^32 or 2^64 scaled, by word width)
    uint64_t mu32;       // floor(2^32/q) (u32 path: fold lazy NTT values back to [0,2q))
    uint64_t inv_last_w;     // inv[1] * N^-1 mod q : last inverse-NTT stage with N^-1 folded in
    uint64_t inv_last_w_sh;  // its Shoup companion
    uint32_t kbits;      // bits(q)
    uint32_t lazy_terms; // how many q^2-bounded products fit the accumulator
};

// Environment switches, read ONCE per context at gpu_context_create (not per launch: a preimage call
// at n = 256 is ~30 launches and each getenv scans the environment under libc's lock).
// gpupoly_reload_env() re-reads them for every live context (tests flip them between calls).
struct EnvSwitches {
    int ntt_path = 0;             // MXX_HIP_NTT_PATH: 0 auto, 1 lds, 2 generic, 3 global
    int ntt14 = 0;                // MXX_HIP_NTT14: 0 grouped signed, 1 whole-vector kernel, 2 grouped unsigned
    bool decompose_fused = true;  // MXX_HIP_DECOMPOSE_FUSED=0 disables digits-in-the-NTT-load
    char matmul_path = 0;         // MXX_HIP_MATMUL_PATH: 0 auto, 'r' reg, 'l' lds, 'd' dma, 'w' dma32 (wide tile), 'm' mfma
    int matmul_tile = 0;          // MXX_HIP_MATMUL_TILE=RCS[p] (tuning: rows, cols, slots per lane of the register tile, p = loads ahead)
    bool gsamp_simple = false;    // MXX_HIP_GSAMP=simple
    bool gsamp_no_uni = false;    // MXX_HIP_GSAMP=nouni: the lane kernel's per-element tower look-up even when chunks are uniform (A/B, tests)
    bool p1_simple = false;       // MXX_HIP_P1=simple
    int sampler_per_lane = 0;     // MXX_HIP_SAMPLER_PER_LANE (0 = sized for one resident round)
    int sampler_fill_every = 0;   // MXX_HIP_SAMPLER_FILL_EVERY = 1..8: keystream refill cadence of the Gaussian lane kernels in checkpoints (0 = per kernel default)
    bool ntt64_int = false;       // MXX_HIP_NTT64=int: 64-bit words keep the integer butterflies (A/B, tests)
    int ntt_phase = 0;            // MXX_HIP_NTT_PHASE: phase mask of the forward 2^14 transform, GPUPOLY_PHASE_TIMING builds only (ntt14.h)
    bool serde_general = false;   // MXX_HIP_SERDE=general: compact store always through the kernels that carry the general Garner path (tests, A/B)
    bool rng_compat = false;      // MXX_HIP_RNG_COMPAT=reference: sample_distribution* keyed exactly as the reference's device RNG (sampling.hip)
    void load();
};

struct GpuContext {
    EnvSwitches env;
    int device = 0;
    std::vector<int> gpu_ids;
    uint32_t logN = 0;
    int N = 0;
    int level = 0;  // top level = limb_count-1
    uint32_t dnum = 0;
    int limb_count = 0;
    bool wide = false;  // true: uint64_t residues
    int word_bytes = 4;
    uint32_t crt_bits = 0;  // max bit width over the moduli
    std::vector<uint64_t> moduli;
    hipStream_t stream = nullptr;
    // device tables
    LimbConst *d_limbs = nullptr;  // [limb_count]
    LimbConst *d_limbs_r = nullptr;  // u32 words: the same with n_inv / inv_last_w (+ companions) multiplied by 2^32 mod q,
                                     // for the inverse transform that multiplies in its load (Montgomery product, ntt14.h)
    void *d_tw_fwd = nullptr;      // [limb][N] words, fwd[bitrev(i)] = psi^i
    void *d_tw_fwd_sh = nullptr;   // Shoup companions
    void *d_tw_inv = nullptr;      // inv[bitrev(i)] = psi^-i
    void *d_tw_inv_sh = nullptr;
    void *d_tw2_fwd = nullptr;     // [limb][N] pairs {-w mod 2^W, Shoup(w)} for the lazy forward kernel
    void *d_tw2_inv = nullptr;     // [limb][N] pairs {w, Shoup(w)} for the lazy inverse kernel
    bool lazy_ok = false;          // every modulus < 2^(W-7): lazy LDS kernels are valid
    bool tight_ok = false;         // 32-bit words, moduli of 26..28 bits: the lazy kernels' TIGHT forms (ntt_lds.h)
    void *d_tw2s_inv = nullptr;    // u32 words, moduli < 2^24: inverse pairs {centred w, floor(w 2^32 / q)} as int32 (ntt14.h)
    bool signed_ok = false;
    // 64-bit words, every modulus below 2^51: the double-precision transforms of ntt_f64.h
    void *d_twf_fwd = nullptr, *d_twf_inv = nullptr;  // [limb][N] {w, w / q} as doubles
    void *d_flimbs = nullptr;                         // [limb] F64Limb
    bool f64_ok = false;
    uint64_t *d_garner = nullptr;  // [limb][limb] : inverse of q_j mod q_i for j<i
    std::vector<uint64_t> garner_inv;  // host copy
    std::vector<LimbConst> limbs;      // host copy
    // bench timer
    hipEvent_t timer_start = nullptr, timer_stop = nullptr;
    std::vector<hipEvent_t> marks;  // lazily created timing marks
    std::mutex mutex;
    bool pool_ok = false;  // cache freed blocks (stream-ordered reuse)
    std::mutex alloc_mutex;
    std::multimap<size_t, void *> free_blocks;      // size -> block
    std::unordered_map<void *, size_t> live_blocks;  // block -> size
    size_t cached_bytes = 0, cache_limit = 0;
    // name of the kernel the product dispatcher launched last on this context (a string literal or a function-local
    // static: bench.py labels its roofline with what actually ran, gpupoly_context_last_kernel)
    std::atomic<const char *> last_kernel{""};
};

struct GpuMatrix {
    GpuContext *ctx = nullptr;
    int level = 0;
    size_t rows = 0, cols = 0;
    int format = GPU_POLY_FORMAT_EVAL;
    void *data = nullptr;  // words [rows*cols][level+1][N]
    size_t bytes = 0;
    bool borrowed = false;  // a row-block view of another matrix (gpupoly_matrix_row_view): data is not freed with it
};

struct GpuEventSet {
    std::vector<hipEvent_t> events;
    int device = 0;
    void *staging = nullptr;  // host-visible staging released after the wait
    GpuContext *ctx = nullptr;
    void *dev_staging = nullptr;
};

struct GpuP1CovarianceCache {
    GpuContext *ctx = nullptr;
    int level = 0;
    size_t d = 0, m = 0, n = 0;
    double sigma = 0, s = 0, dgg_stddev = 0;
    double *sqrt_var = nullptr;      // [coeff][row]
    double *update_coeff = nullptr;  // [coeff][sampled_row][updated_row]
    void *karney_div = nullptr;      // [coeff][row] KarneyDivisor of sqrt_var (rng.h)
};

// ---- one thread per item -----------------------------------------------------------------------
// HIP rejects launches with gridDim.x * blockDim.x >= 2^32 ("invalid configuration argument"), which a
// 64x1024 gadget matrix at n = 2^14 already exceeds: such kernels use item_grid() / item_index().
static inline dim3 item_grid(size_t items, unsigned threads) {
    const size_t blocks = (items + threads - 1) / threads;
    const size_t max_x = (static_cast<size_t>(1) << 31) / threads;
    if (blocks <= max_x) return dim3(static_cast<unsigned>(blocks ? blocks : 1));
    const size_t y = (blocks + max_x - 1) / max_x;
    return dim3(static_cast<unsigned>((blocks + y - 1) / y), static_cast<unsigned>(y));
}
#if defined(__HIPCC__)
__device__ __forceinline__ size_t item_index() {
    return (static_cast<size_t>(blockIdx.y) * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
}
#endif

// every kernel launch of the library goes through this (bench.py prints launches per step for the launch-bound
// small-ring chain: gpupoly_launch_count)
extern std::atomic<uint64_t> g_kernel_launches;
// Launch trace (gpupoly_trace_begin / gpupoly_trace_end, runtime.hip): while it is on, every launch is bracketed by two
// hipEvents on the stream it is launched on and recorded with its kernel's name, grid and - where the launcher states
// them with MXX_TRACE_BYTES - the algorithmic bytes of its operands (each read once + written once).  bench.py composes
// the roofline of a multi-kernel call (a preimage, a chain step) from it.  Off: one relaxed load per launch.
extern std::atomic<int> g_trace_on;
void trace_launch_begin(const char *name, hipStream_t stream, dim3 grid, dim3 block);
void trace_launch_end(hipStream_t stream);
void trace_set_bytes(double bytes);
#define MXX_TRACE_BYTES(b)                                                                        \
    do {                                                                                          \
        if (g_trace_on.load(std::memory_order_relaxed)) trace_set_bytes(static_cast<double>(b)); \
    } while (0)
#define MXX_LAUNCH(kern, grid, block, lds, strm, ...)                          \
    do {                                                                       \
        g_kernel_launches.fetch_add(1, std::memory_order_relaxed);             \
        const bool mxx_tr_ = g_trace_on.load(std::memory_order_relaxed) != 0;  \
        if (mxx_tr_) trace_launch_begin(#kern, strm, grid, block);             \
        hipLaunchKernelGGL(kern, grid, block, lds, strm, __VA_ARGS__);         \
        if (mxx_tr_) trace_launch_end(strm);                                   \
    } while (0)
// a runtime copy / memset on the context's stream, traced like a launch (the statement is the HIP_TRY'd call)
#define MXX_TRACED_COPY(name, strm, bytes, stmt)                               \
    do {                                                                       \
        const bool mxx_tr_ = g_trace_on.load(std::memory_order_relaxed) != 0;  \
        if (mxx_tr_) {                                                         \
            trace_set_bytes(static_cast<double>(bytes));                       \
            trace_launch_begin(name, strm, dim3(0), dim3(0));                  \
        }                                                                      \
        stmt;                                                                  \
        if (mxx_tr_) trace_launch_end(strm);                                   \
    } while (0)

// ---- error plumbing ---------------------------------------------------------
int set_error(const char *msg);
int set_error(const std::string &msg);
int set_error(hipError_t err, const char *what);

#define HIP_TRY(expr)                                          \
    do {                                                       \
        hipError_t _e = (expr);                                \
        if (_e != hipSuccess) return set_error(_e, #expr);     \
    } while (0)

// roctx range around an ABI entry (rocprofv3 --marker-trace attributes kernels to the FFI call that launched them).
// Active under rocprofv3 (ROCP_TOOL_LIBRARIES is set by it) or with MXX_HIP_ROCTX=1; one predictable branch otherwise.
extern int (*g_roctx_push)(const char *);
extern int (*g_roctx_pop)();
void roctx_init_once();
struct RoctxScope {
    bool on;
    explicit RoctxScope(const char *name) {
        roctx_init_once();
        on = g_roctx_push != nullptr;
        if (on) g_roctx_push(name);
    }
    ~RoctxScope() {
        if (on) g_roctx_pop();
    }
};

// every extern "C" body is wrapped so no exception crosses the ABI
#define ABI_GUARD_BEGIN try { RoctxScope roctx_scope_(__func__);
#define ABI_GUARD_END                                                         \
    }                                                                         \
    catch (const std::exception &e) { return set_error(e.what()); }           \
    catch (...) { return set_error("unknown exception in libgpupoly"); }

// ---- helpers ------------------------------------------------------------------
inline size_t matrix_polys(const GpuMatrix *m) { return m->rows * m->cols; }
inline size_t matrix_limbs(const GpuMatrix *m) { return static_cast<size_t>(m->level) + 1; }
inline size_t matrix_words(const GpuMatrix *m) { return matrix_polys(m) * matrix_limbs(m) * static_cast<size_t>(m->ctx->N); }

// sampling.hip: fills `out` with samples; keep_coeff leaves them in the coefficient domain
int sample_impl(GpuMatrix *out, int dist, double sigma, GpuRngSeed seed, size_t full_ncol, size_t col_offset, bool keep_coeff);

int ctx_activate(const GpuContext *ctx);                   // hipSetDevice
bool ctx_is_registered(const GpuContext *ctx);             // still a live context of this process (runtime.hip's registry)
int ctx_alloc(GpuContext *ctx, size_t bytes, void **out);  // stream-ordered
void ctx_free(GpuContext *ctx, void *ptr);                 // stream-ordered
// a stream-ordered block that goes back to the context's cache when the scope ends (error paths
// included) unless ownership is handed on with release()
struct CtxBlock {
    GpuContext *ctx;
    void *ptr = nullptr;
    explicit CtxBlock(GpuContext *c) : ctx(c) {}
    CtxBlock(const CtxBlock &) = delete;
    CtxBlock &operator=(const CtxBlock &) = delete;
    ~CtxBlock() {
        if (ptr) ctx_free(ctx, ptr);
    }
    int alloc(size_t bytes) { return ctx_alloc(ctx, bytes, &ptr); }
    void *release() {
        void *p = ptr;
        ptr = nullptr;
        return p;
    }
};
int matrix_check_same_shape(const GpuMatrix *a, const GpuMatrix *b, const char *who);

// internal launchers shared across translation units
int launch_ntt(GpuContext *ctx, void *data, size_t vectors, int limbs_per_poly, bool inverse);
// tuned LDS kernels (ntt_lds_u32.hip / ntt_lds_u64.hip); return -1 when no tuned kernel covers logN
int launch_ntt_lds_u32(GpuContext *ctx, uint32_t *data, size_t vectors, uint32_t L, bool inverse);
int launch_ntt_digits_u32(GpuContext *ctx, uint32_t *out, const uint32_t *coeff, size_t out_vectors, uint32_t L,
                          uint32_t src_cols, uint32_t towers, uint32_t dpt, uint32_t base_bits, size_t k);
int launch_ntt_lds_u64(GpuContext *ctx, uint64_t *data, size_t vectors, uint32_t L, bool inverse);
int launch_ntt_digits_u64(GpuContext *ctx, uint64_t *out, const uint64_t *coeff, size_t out_vectors, uint32_t L,
                          uint32_t src_cols, uint32_t towers, uint32_t dpt, uint32_t base_bits, size_t k);
// out <- INTT(in o w), w one resident EVAL-form ring element [L][N]; -1: no fused kernel for this context
int launch_intt_oop_u32(GpuContext *ctx, uint32_t *out, const uint32_t *in, size_t vectors, uint32_t L);
int launch_ntt_add_u32(GpuContext *ctx, uint32_t *out, const uint32_t *src, const uint32_t *add, size_t vectors, uint32_t L);
int launch_mul_intt_u32(GpuContext *ctx, uint32_t *out, const uint32_t *in, const uint32_t *w, size_t vectors, uint32_t L);
int launch_matmul(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);
int launch_matmul_dma_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);  // -1: shape not supported
int launch_matmul_dma32_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);  // 32 slots x 32x32 tile, 16 waves
int launch_matmul_mfma_u32(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);  // -1: shape / moduli not supported
int launch_copy_block(GpuMatrix *out, const GpuMatrix *src, size_t dst_row, size_t dst_col, size_t src_row,
                      size_t src_col, size_t rows, size_t cols, bool add);
int launch_scatter_i64(GpuMatrix *out, const int64_t *vals);  // int64 [poly][N] -> residues in every limb
