// runtime.hip — context, per-prime constants, events, device queries, errors.
// Replaces cuda/src/Runtime.cu of the reference behind the same C ABI
// (cuda/include/Runtime.cuh:20-44,108), re-designed for one HIP stream per
// context and the OpenFHE NTT convention.
#include "common.h"

#include <algorithm>
#include <cstdlib>
#include <iterator>
#include <map>
#include <cstring>
#include <dlfcn.h>

typedef unsigned __int128 u128h;

// ---- thread-local error string (Runtime.cu:16-27,817-820 behaviour) -----------
static thread_local std::string g_last_error;

int set_error(const char *msg) {
    g_last_error = msg ? msg : "unknown error";
    return 1;
}
int set_error(const std::string &msg) {
    g_last_error = msg;
    return 1;
}
int set_error(hipError_t err, const char *what) {
    g_last_error = std::string(what ? what : "hip call") + ": " + hipGetErrorString(err);
    (void)hipGetLastError();
    return 1;
}

extern "C" const char *gpu_last_error(void) { return g_last_error.c_str(); }
extern "C" int gpu_set_last_error(const char *msg) { return set_error(msg); }
std::atomic<uint64_t> g_kernel_launches{0};
// kernel launches issued by this library since it was loaded (all contexts; copies and memsets not included)
extern "C" uint64_t gpupoly_launch_count(void) { return g_kernel_launches.load(std::memory_order_relaxed); }
extern "C" const char *gpupoly_version(void) { return "gpupoly-mi355x 0.1 (gfx950)"; }

// ---- launch trace (extension; common.h: MXX_LAUNCH) ----------------------------------------------------------
std::atomic<int> g_trace_on{0};
namespace {
struct TraceEntry {
    const char *name;
    hipEvent_t e0, e1;
    unsigned long long blocks;
    unsigned threads;
    double bytes;
    int device;
    bool closed;
};
std::mutex g_trace_mutex;
std::vector<TraceEntry> g_trace;
// events of earlier traces, reused - one pool per device: an event belongs to the device that was current when it was
// created, and recording it on another device's stream fails with an invalid handle
std::map<int, std::vector<hipEvent_t>> g_trace_pool;
std::string g_trace_report;
thread_local double t_trace_bytes = 0;
thread_local long t_trace_open = -1;
hipEvent_t trace_event(int device) {
    std::vector<hipEvent_t> &pool = g_trace_pool[device];
    if (!pool.empty()) {
        hipEvent_t e = pool.back();
        pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
        (void)hipGetLastError();  // the caller's own hipGetLastError() after its launch must not see this one
        return nullptr;
    }
    return e;
}
void trace_event_return(int device, hipEvent_t e) {
    if (e) g_trace_pool[device].push_back(e);
}
}  // namespace

void trace_set_bytes(double bytes) { t_trace_bytes = bytes; }

void trace_launch_begin(const char *name, hipStream_t stream, dim3 grid, dim3 block) {
    std::lock_guard<std::mutex> lk(g_trace_mutex);
    int device = 0;
    (void)hipGetDevice(&device);  // the launcher activated the stream's device before calling
    TraceEntry en{name, trace_event(device), trace_event(device), 1ull * grid.x * grid.y * grid.z, block.x * block.y * block.z,
                  t_trace_bytes, device, false};
    t_trace_bytes = 0;
    t_trace_open = -1;
    if (!en.e0 || !en.e1 || hipEventRecord(en.e0, stream) != hipSuccess) {
        // a launch that cannot be traced is still launched: hand the events back and leave no HIP error behind for the
        // launcher's HIP_TRY(hipGetLastError()) to trip over
        (void)hipGetLastError();
        trace_event_return(device, en.e0);
        trace_event_return(device, en.e1);
        return;
    }
    g_trace.push_back(en);
    t_trace_open = static_cast<long>(g_trace.size()) - 1;
}

void trace_launch_end(hipStream_t stream) {
    std::lock_guard<std::mutex> lk(g_trace_mutex);
    if (t_trace_open < 0 || static_cast<size_t>(t_trace_open) >= g_trace.size()) return;
    TraceEntry &en = g_trace[static_cast<size_t>(t_trace_open)];
    en.closed = hipEventRecord(en.e1, stream) == hipSuccess;
    if (!en.closed) (void)hipGetLastError();
    t_trace_open = -1;
}

// A one-thread kernel whose only purpose is its name in a profiler's dispatch list: bench.py launches one on each side of
// its timed region, and tools/pmc_window.py sums rocprofv3's per-dispatch counters BETWEEN the two - the set-up's launches
// of the same kernels (other sizes) stay out of the per-launch figures.
__global__ void gpupoly_marker_kernel(uint32_t id, uint32_t *sink) {
    if (sink && id == 0xffffffffu) *sink = id;  // never true for the ids bench.py uses: the kernel has no effect
}
extern "C" int gpupoly_marker_launch(GpuContext *ctx, uint32_t id) {
    ABI_GUARD_BEGIN
    if (!ctx) return set_error("gpupoly_marker_launch: null ctx");
    if (ctx_activate(ctx)) return 1;
    hipLaunchKernelGGL(gpupoly_marker_kernel, dim3(1), dim3(1), 0, ctx->stream, id & 0x7fffffffu, static_cast<uint32_t *>(nullptr));
    HIP_TRY(hipGetLastError());
    return 0;
    ABI_GUARD_END
}

// Start recording every launch of this library (all contexts, all threads).  Entries accumulate until gpupoly_trace_end.
extern "C" int gpupoly_trace_begin(void) {
    ABI_GUARD_BEGIN
    std::lock_guard<std::mutex> lk(g_trace_mutex);
    for (const TraceEntry &en : g_trace) {
        trace_event_return(en.device, en.e0);
        trace_event_return(en.device, en.e1);
    }
    g_trace.clear();
    g_trace_on.store(1, std::memory_order_relaxed);
    return 0;
    ABI_GUARD_END
}

// Stop recording, wait for the recorded launches and return one line per launch, in launch order:
// "name \t blocks \t threads per block \t algorithmic bytes (0: not stated) \t milliseconds \n".  The string belongs to the
// library and stays valid until the next gpupoly_trace_begin / gpupoly_trace_end; NULL on error (gpu_last_error()).
extern "C" const char *gpupoly_trace_end(void) {
    try {
        g_trace_on.store(0, std::memory_order_relaxed);
        std::lock_guard<std::mutex> lk(g_trace_mutex);
        g_trace_report.clear();
        int prev = 0;
        (void)hipGetDevice(&prev);
        char line[512];
        for (const TraceEntry &en : g_trace) {
            if (!en.closed) continue;
            (void)hipSetDevice(en.device);
            float ms = 0;
            hipError_t e = hipEventSynchronize(en.e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, en.e0, en.e1);
            if (e != hipSuccess) {
                (void)hipSetDevice(prev);
                set_error(e, "gpupoly_trace_end: hipEventElapsedTime");
                return nullptr;
            }
            std::snprintf(line, sizeof line, "%s\t%llu\t%u\t%.0f\t%.6f\n", en.name, en.blocks, en.threads, en.bytes, ms);
            g_trace_report += line;
        }
        for (const TraceEntry &en : g_trace) {  // the events go back to the pool: a second _end reports nothing
            trace_event_return(en.device, en.e0);
            trace_event_return(en.device, en.e1);
        }
        g_trace.clear();
        (void)hipSetDevice(prev);
        return g_trace_report.c_str();
    } catch (...) {
        set_error("gpupoly_trace_end: exception");
        return nullptr;
    }
}

// ---- roctx (tracing) ---------------------------------------------------------------
int (*g_roctx_push)(const char *) = nullptr;
int (*g_roctx_pop)() = nullptr;
void roctx_init_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char *sw = std::getenv("MXX_HIP_ROCTX");
        const bool profiled = std::getenv("ROCP_TOOL_LIBRARIES") != nullptr || std::getenv("ROCPROFILER_REGISTER_FORCE_LOAD") != nullptr;
        if (sw ? sw[0] == '0' : !profiled) return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (!h) continue;
            auto push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            auto pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) {
                g_roctx_pop = pop;
                g_roctx_push = push;
                return;
            }
        }
    });
}

// ---- host number theory ---------------------------------------------------------
static inline uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128h)a * b) % q); }
static uint64_t h_powmod(uint64_t b, uint64_t e, uint64_t q) {
    uint64_t r = 1 % q;
    b %= q;
    while (e) {
        if (e & 1) r = h_mulmod(r, b, q);
        b = h_mulmod(b, b, q);
        e >>= 1;
    }
    return r;
}
static uint64_t h_invmod_prime(uint64_t a, uint64_t q) { return h_powmod(a % q, q - 2, q); }

static bool h_is_prime(uint64_t n) {
    static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (uint64_t b : bases)
        if (n % b == 0) return n == b;
    uint64_t d = n - 1;
    int s = 0;
    while (!(d & 1)) { d >>= 1; ++s; }
    for (uint64_t b : bases) {
        uint64_t x = h_powmod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool comp = true;
        for (int r = 1; r < s; ++r) {
            x = h_mulmod(x, x, n);
            if (x == n - 1) { comp = false; break; }
        }
        if (comp) return false;
    }
    return true;
}

// psi = MIN over all primitive 2N-th roots of unity mod q — OpenFHE RootOfUnity(),
// restated in-tree by the reference at src/gadgets/ntt/mod.rs:96-128.
static uint64_t h_min_primitive_root(uint64_t q, uint64_t order) {
    uint64_t qm1 = q - 1;
    int v = __builtin_ctzll(qm1);
    int want = __builtin_ctzll(order);
    uint64_t odd = qm1 >> v;
    uint64_t maximal = 0;
    for (uint64_t x = 2; x < q; ++x) {
        uint64_t r = h_powmod(x, odd, q);
        if (h_powmod(r, 1ull << (v - 1), q) != 1) { maximal = r; break; }
    }
    uint64_t root = h_powmod(maximal, 1ull << (v - want), q);
    uint64_t sq = h_mulmod(root, root, q);
    uint64_t cur = root, best = root;
    for (uint64_t i = 1; i < order / 2; ++i) {
        cur = h_mulmod(cur, sq, q);
        if (cur < best) best = cur;
    }
    return best;
}

static inline uint32_t h_bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0;
    for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}
static inline uint32_t h_bits(uint64_t v) { return v ? 64 - (uint32_t)__builtin_clzll(v) : 0; }

// ---- environment switches + registry of live contexts -----------------------------------
void EnvSwitches::load() {
    *this = EnvSwitches();
    if (const char *e = std::getenv("MXX_HIP_NTT_PATH")) {  // lds | generic | global
        if (e[0] == 'l') ntt_path = 1;
        else if (e[0] == 'g' && e[1] == 'e') ntt_path = 2;
        else if (e[0] == 'g' && e[1] == 'l') ntt_path = 3;
    }
    if (const char *e = std::getenv("MXX_HIP_NTT14")) ntt14 = e[0] == 'w' ? 1 : (e[0] == 'u' ? 2 : 0);
    if (const char *e = std::getenv("MXX_HIP_DECOMPOSE_FUSED")) decompose_fused = e[0] != '0';
    if (const char *e = std::getenv("MXX_HIP_MATMUL_PATH")) matmul_path = e[0];
    if (const char *e = std::getenv("MXX_HIP_MATMUL_TILE")) {
        if (e[0] && e[1] && e[2]) matmul_tile = ((e[0] - '0') * 100 + (e[1] - '0') * 10 + (e[2] - '0')) * 2 + (e[3] == 'p');
    }
    if (const char *e = std::getenv("MXX_HIP_GSAMP")) {
        gsamp_simple = e[0] == 's';
        gsamp_no_uni = e[0] == 'n';
    }
    if (const char *e = std::getenv("MXX_HIP_P1")) p1_simple = e[0] == 's';
    if (const char *e = std::getenv("MXX_HIP_NTT64")) ntt64_int = e[0] == 'i';
    if (const char *e = std::getenv("MXX_HIP_RNG_COMPAT")) rng_compat = e[0] == 'r';
    if (const char *e = std::getenv("MXX_HIP_SERDE")) serde_general = e[0] == 'g';
    if (const char *e = std::getenv("MXX_HIP_SAMPLER_PER_LANE")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 4096) sampler_per_lane = v;
    }
    if (const char *e = std::getenv("MXX_HIP_NTT_PHASE")) {
        const int v = std::atoi(e);
        if (v >= 0 && v <= 127) ntt_phase = v;
    }
    if (const char *e = std::getenv("MXX_HIP_SAMPLER_FILL_EVERY")) {
        const int v = std::atoi(e);
        if (v >= 1 && v <= 8) sampler_fill_every = v;
    }
}

static std::mutex g_registry_mutex;
static std::vector<GpuContext *> g_contexts;  // every live context of the process
bool ctx_is_registered(const GpuContext *ctx) {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    return std::find(g_contexts.begin(), g_contexts.end(), ctx) != g_contexts.end();
}

extern "C" int gpupoly_reload_env(void) {
    std::lock_guard<std::mutex> lk(g_registry_mutex);
    for (GpuContext *c : g_contexts) c->env.load();
    return 0;
}

// ---- context ----------------------------------------------------------------------
int ctx_activate(const GpuContext *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

// Stream-ordered caching allocator.  Every operation of a context is enqueued on its one
// stream, so a block returned here can be handed out again immediately: whatever uses it
// next is ordered behind whatever used it last.  Blocks come from plain hipMalloc (the
// runtime's hipMallocAsync pool corrupted large D2D/D2H copies under rapid reuse on ROCm
// 7.2 / gfx950 — tools/stress_prims.py — so it is not used); cached bytes are capped by
// MXX_HIP_MEMPOOL_RELEASE_THRESHOLD_BYTES (the reference's
// MXX_CUDA_MEMPOOL_RELEASE_THRESHOLD_BYTES is honoured too, Runtime.cu:337-361).
static size_t round_block(size_t bytes) {
    const size_t gran = bytes >= (size_t(1) << 20) ? (size_t(1) << 20) : 4096;
    return (bytes + gran - 1) / gran * gran;
}

static void cache_trim(GpuContext *ctx, size_t limit) {
    // caller holds ctx->alloc_mutex; hipFree synchronises, which is fine on this rare path
    while (ctx->cached_bytes > limit && !ctx->free_blocks.empty()) {
        auto it = std::prev(ctx->free_blocks.end());  // largest first
        (void)hipFree(it->second);
        ctx->cached_bytes -= it->first;
        ctx->free_blocks.erase(it);
    }
}

int ctx_alloc(GpuContext *ctx, size_t bytes, void **out) {
    *out = nullptr;
    if (bytes == 0) return 0;
    const size_t want = round_block(bytes);
    std::lock_guard<std::mutex> lk(ctx->alloc_mutex);
    auto it = ctx->free_blocks.lower_bound(want);
    if (it != ctx->free_blocks.end() && it->first <= want + want / 4) {
        *out = it->second;
        ctx->live_blocks[*out] = it->first;
        ctx->cached_bytes -= it->first;
        ctx->free_blocks.erase(it);
        return 0;
    }
    hipError_t e = hipMalloc(out, want);
    if (e != hipSuccess) {  // out of memory: drop this context's cache and retry
        (void)hipGetLastError();
        cache_trim(ctx, 0);
        e = hipMalloc(out, want);
    }
    if (e != hipSuccess) {
        // still out of memory: other contexts on the same device (tests create many; bench N>1 shares
        // the card with torch / RCCL) may be sitting on cached blocks - drop those too.  try_lock, so
        // two contexts trimming each other cannot deadlock (we hold our own alloc_mutex).
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> reg(g_registry_mutex);
            for (GpuContext *other : g_contexts) {
                if (other == ctx || other->device != ctx->device) continue;
                if (!other->alloc_mutex.try_lock()) continue;
                cache_trim(other, 0);
                other->alloc_mutex.unlock();
            }
        }
        e = hipMalloc(out, want);
        if (e != hipSuccess) {
            *out = nullptr;
            return set_error(e, "hipMalloc");
        }
    }
    ctx->live_blocks[*out] = want;
    return 0;
}

void ctx_free(GpuContext *ctx, void *ptr) {
    if (!ptr) return;
    std::lock_guard<std::mutex> lk(ctx->alloc_mutex);
    auto it = ctx->live_blocks.find(ptr);
    if (it == ctx->live_blocks.end()) {
        (void)hipFree(ptr);
        return;
    }
    const size_t size = it->second;
    ctx->live_blocks.erase(it);
    if (!ctx->pool_ok) {  // MXX_HIP_DISABLE_MEMPOOL=1: no caching
        (void)hipFree(ptr);
        return;
    }
    ctx->free_blocks.emplace(size, ptr);
    ctx->cached_bytes += size;
    cache_trim(ctx, ctx->cache_limit);
}

template <typename W>
static int upload_tables(GpuContext *ctx, const std::vector<std::vector<uint64_t>> &fwd,
                         const std::vector<std::vector<uint64_t>> &inv) {
    const size_t N = ctx->N, L = ctx->limb_count;
    const unsigned shift = sizeof(W) * 8;
    std::vector<W> h_f(L * N), h_fs(L * N), h_i(L * N), h_is(L * N);
    for (size_t l = 0; l < L; ++l) {
        uint64_t q = ctx->moduli[l];
        for (size_t j = 0; j < N; ++j) {
            h_f[l * N + j] = (W)fwd[l][j];
            h_fs[l * N + j] = (W)(((u128h)fwd[l][j] << shift) / q);
            h_i[l * N + j] = (W)inv[l][j];
            h_is[l * N + j] = (W)(((u128h)inv[l][j] << shift) / q);
        }
    }
    size_t bytes = L * N * sizeof(W);
    HIP_TRY(hipMalloc(&ctx->d_tw_fwd, bytes));
    HIP_TRY(hipMalloc(&ctx->d_tw_fwd_sh, bytes));
    HIP_TRY(hipMalloc(&ctx->d_tw_inv, bytes));
    HIP_TRY(hipMalloc(&ctx->d_tw_inv_sh, bytes));
    HIP_TRY(hipMemcpy(ctx->d_tw_fwd, h_f.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tw_fwd_sh, h_fs.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tw_inv, h_i.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tw_inv_sh, h_is.data(), bytes, hipMemcpyHostToDevice));
    // interleaved {twiddle, companion} pairs for the lazy LDS kernels; forward twiddles negated
    std::vector<W> h_pf(2 * L * N), h_pi(2 * L * N);
    for (size_t i = 0; i < L * N; ++i) {
        h_pf[2 * i] = static_cast<W>(0) - h_f[i];
        h_pf[2 * i + 1] = h_fs[i];
        h_pi[2 * i] = h_i[i];
        h_pi[2 * i + 1] = h_is[i];
    }
    HIP_TRY(hipMalloc(&ctx->d_tw2_fwd, 2 * bytes));
    HIP_TRY(hipMalloc(&ctx->d_tw2_inv, 2 * bytes));
    HIP_TRY(hipMemcpy(ctx->d_tw2_fwd, h_pf.data(), 2 * bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tw2_inv, h_pi.data(), 2 * bytes, hipMemcpyHostToDevice));
    if (sizeof(W) == 8 && ctx->crt_bits <= 51 && N >= 2) {
        // double-precision tables for ntt_f64.h: {w, w / q} (the quotient of two exactly representable integers is
        // correctly rounded, so the device sees the same bits on every host)
        struct HostTwF { double w, wi; };
        struct HostF64Limb { double q, qinv; HostTwF n_inv, last_w; };
        std::vector<HostTwF> tf(L * N), ti(L * N);
        std::vector<HostF64Limb> fl(L);
        for (size_t l = 0; l < L; ++l) {
            const double qd = static_cast<double>(ctx->moduli[l]);
            for (size_t j = 0; j < N; ++j) {
                tf[l * N + j] = {static_cast<double>(fwd[l][j]), static_cast<double>(fwd[l][j]) / qd};
                ti[l * N + j] = {static_cast<double>(inv[l][j]), static_cast<double>(inv[l][j]) / qd};
            }
            const LimbConst &lc = ctx->limbs[l];
            fl[l] = {qd, 1.0 / qd, {static_cast<double>(lc.n_inv), static_cast<double>(lc.n_inv) / qd},
                     {static_cast<double>(lc.inv_last_w), static_cast<double>(lc.inv_last_w) / qd}};
        }
        HIP_TRY(hipMalloc(&ctx->d_twf_fwd, sizeof(HostTwF) * L * N));
        HIP_TRY(hipMalloc(&ctx->d_twf_inv, sizeof(HostTwF) * L * N));
        HIP_TRY(hipMalloc(&ctx->d_flimbs, sizeof(HostF64Limb) * L));
        HIP_TRY(hipMemcpy(ctx->d_twf_fwd, tf.data(), sizeof(HostTwF) * L * N, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_twf_inv, ti.data(), sizeof(HostTwF) * L * N, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->d_flimbs, fl.data(), sizeof(HostF64Limb) * L, hipMemcpyHostToDevice));
        ctx->f64_ok = true;
    }
    if (ctx->signed_ok && sizeof(W) == 4) {
        // centred twiddle w^ in (-q/2, q/2] and w^' = floor(w^ * 2^32 / q), both as int32 bit patterns
        auto signed_pair = [](uint64_t w, uint64_t q, W *dst) {
            const int64_t c = w > q / 2 ? static_cast<int64_t>(w) - static_cast<int64_t>(q) : static_cast<int64_t>(w);
            __int128 num = static_cast<__int128>(c) << 32;
            __int128 fl = num / static_cast<__int128>(q);
            if (num % static_cast<__int128>(q) < 0) fl -= 1;  // floor for negatives
            dst[0] = static_cast<W>(static_cast<uint32_t>(static_cast<int32_t>(c)));
            dst[1] = static_cast<W>(static_cast<uint32_t>(static_cast<int32_t>(fl)));
        };
        for (size_t l = 0; l < L; ++l)
            for (size_t j = 0; j < N; ++j) signed_pair(inv[l][j], ctx->moduli[l], &h_pi[2 * (l * N + j)]);
        HIP_TRY(hipMalloc(&ctx->d_tw2s_inv, 2 * bytes));
        HIP_TRY(hipMemcpy(ctx->d_tw2s_inv, h_pi.data(), 2 * bytes, hipMemcpyHostToDevice));
    }
    return 0;
}

static void context_release(GpuContext *ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> reg(g_registry_mutex);
        g_contexts.erase(std::remove(g_contexts.begin(), g_contexts.end(), ctx), g_contexts.end());
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &kv : ctx->free_blocks) (void)hipFree(kv.second);
    for (auto &kv : ctx->live_blocks) (void)hipFree(kv.first);
    if (ctx->d_limbs) (void)hipFree(ctx->d_limbs);
    if (ctx->d_limbs_r) (void)hipFree(ctx->d_limbs_r);
    if (ctx->d_twf_fwd) (void)hipFree(ctx->d_twf_fwd);
    if (ctx->d_twf_inv) (void)hipFree(ctx->d_twf_inv);
    if (ctx->d_flimbs) (void)hipFree(ctx->d_flimbs);
    if (ctx->d_tw_fwd) (void)hipFree(ctx->d_tw_fwd);
    if (ctx->d_tw_fwd_sh) (void)hipFree(ctx->d_tw_fwd_sh);
    if (ctx->d_tw_inv) (void)hipFree(ctx->d_tw_inv);
    if (ctx->d_tw_inv_sh) (void)hipFree(ctx->d_tw_inv_sh);
    if (ctx->d_garner) (void)hipFree(ctx->d_garner);
    if (ctx->d_tw2_fwd) (void)hipFree(ctx->d_tw2_fwd);
    if (ctx->d_tw2_inv) (void)hipFree(ctx->d_tw2_inv);
    if (ctx->d_tw2s_inv) (void)hipFree(ctx->d_tw2s_inv);
    if (ctx->timer_start) (void)hipEventDestroy(ctx->timer_start);
    if (ctx->timer_stop) (void)hipEventDestroy(ctx->timer_stop);
    for (hipEvent_t ev : ctx->marks)
        if (ev) (void)hipEventDestroy(ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int gpu_context_create(uint32_t logN, uint32_t L, uint32_t dnum, const uint64_t *moduli,
                                  size_t moduli_len, const int *gpu_ids, size_t gpu_ids_len, GpuContext **out_ctx) {
    ABI_GUARD_BEGIN
    if (!out_ctx) return set_error("gpu_context_create: null out_ctx");
    *out_ctx = nullptr;
    if (!moduli || moduli_len == 0) return set_error("gpu_context_create: empty moduli");
    if (moduli_len != static_cast<size_t>(L) + 1) return set_error("gpu_context_create: moduli_len must equal L+1");
    if (moduli_len > GPUPOLY_MAX_LIMBS) return set_error("gpu_context_create: too many limbs");
    if (!gpu_ids || gpu_ids_len == 0) return set_error("gpu_context_create: empty gpu_ids");
    if (logN < 1 || logN > 17) return set_error("gpu_context_create: logN out of range [1,17]");
    int dev_count = 0;
    HIP_TRY(hipGetDeviceCount(&dev_count));
    for (size_t i = 0; i < gpu_ids_len; ++i) {
        if (gpu_ids[i] < 0 || gpu_ids[i] >= dev_count) return set_error("gpu_context_create: invalid gpu id");
        for (size_t j = 0; j < i; ++j)
            if (gpu_ids[j] == gpu_ids[i]) return set_error("gpu_context_create: duplicate gpu id");
    }
    const uint64_t N = 1ull << logN;
    bool wide = false;
    for (size_t i = 0; i < moduli_len; ++i) {
        uint64_t q = moduli[i];
        if (q < 3 || (q - 1) % (2 * N) != 0) return set_error("gpu_context_create: modulus must be = 1 (mod 2N)");
        if (!h_is_prime(q)) return set_error("gpu_context_create: modulus is not prime");
        if (q >> 62) return set_error("gpu_context_create: modulus must be < 2^62");
        if (q >> 31) wide = true;
        for (size_t j = 0; j < i; ++j)
            if (moduli[j] == q) return set_error("gpu_context_create: duplicate modulus");
    }

    GpuContext *ctx = new GpuContext();
    ctx->device = gpu_ids[0];
    ctx->gpu_ids.assign(gpu_ids, gpu_ids + gpu_ids_len);
    ctx->logN = logN;
    ctx->N = static_cast<int>(N);
    ctx->level = static_cast<int>(L);
    ctx->dnum = dnum;
    ctx->limb_count = static_cast<int>(moduli_len);
    ctx->wide = wide;
    ctx->word_bytes = wide ? 8 : 4;
    ctx->moduli.assign(moduli, moduli + moduli_len);

    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) { delete ctx; return set_error(e, "hipSetDevice"); }
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return set_error(e, "hipStreamCreate"); }

    ctx->env.load();
    {
        ctx->pool_ok = true;
        // default cap on cached (freed, reusable) bytes: half of what the device has free now, so that
        // several contexts, torch and RCCL on one card do not starve each other; an allocation that
        // still fails trims every context's cache on the device before giving up (ctx_alloc)
        size_t free_b = 0, total_b = 0;
        ctx->cache_limit = size_t(64) << 30;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b) ctx->cache_limit = free_b / 2;
        else (void)hipGetLastError();
        const char *env = std::getenv("MXX_HIP_MEMPOOL_RELEASE_THRESHOLD_BYTES");
        if (!env) env = std::getenv("MXX_CUDA_MEMPOOL_RELEASE_THRESHOLD_BYTES");
        if (env && *env) ctx->cache_limit = std::strtoull(env, nullptr, 10);
        const char *nopool = std::getenv("MXX_HIP_DISABLE_MEMPOOL");
        if (nopool && *nopool == '1') ctx->pool_ok = false;
    }

    // per-prime constants
    std::vector<std::vector<uint64_t>> fwd(moduli_len), inv(moduli_len);
    ctx->limbs.resize(moduli_len);
    uint32_t crt_bits = 0;
    for (size_t l = 0; l < moduli_len; ++l) {
        uint64_t q = moduli[l];
        uint64_t psi = h_min_primitive_root(q, 2 * N);
        uint64_t ipsi = h_invmod_prime(psi, q);
        fwd[l].resize(N);
        inv[l].resize(N);
        uint64_t p = 1, ip = 1;
        for (uint64_t i = 0; i < N; ++i) {
            uint32_t r = h_bitrev(static_cast<uint32_t>(i), logN);
            fwd[l][r] = p;
            inv[l][r] = ip;
            p = h_mulmod(p, psi, q);
            ip = h_mulmod(ip, ipsi, q);
        }
        LimbConst &lc = ctx->limbs[l];
        lc.q = q;
        lc.kbits = h_bits(q);
        crt_bits = std::max(crt_bits, lc.kbits);
        lc.mu = static_cast<uint64_t>((((u128h)1) << (2 * lc.kbits)) / q);
        lc.mu64 = static_cast<uint64_t>((((u128h)1) << 64) / q);
        lc.n_inv = h_invmod_prime(N % q, q);
        lc.n_inv_sh = static_cast<uint64_t>(((u128h)lc.n_inv << (wide ? 64 : 32)) / q);
        lc.mu32 = (((uint64_t)1) << 32) / q;
        lc.inv_last_w = h_mulmod(inv[l][1 % N], lc.n_inv, q);
        lc.inv_last_w_sh = static_cast<uint64_t>(((u128h)lc.inv_last_w << (wide ? 64 : 32)) / q);
        if (!wide) {
            // the accumulator carries a folded residue (< q) into every window of products:
            // (q - 1) + terms * (q - 1)^2 must stay below 2^64
            u128h q2 = (u128h)(q - 1) * (q - 1);
            u128h terms = ((((u128h)1) << 64) - q) / q2;
            lc.lazy_terms = terms > (1u << 20) ? (1u << 20) : static_cast<uint32_t>(terms);
        } else {
            // 128-bit accumulators: (q - 1) + terms * (q - 1)^2 < 2^128 (q < 2^62: at least 15 terms)
            u128h q2 = (u128h)(q - 1) * (q - 1);
            u128h terms = (~(u128h)0 - q) / q2;
            lc.lazy_terms = terms > (1u << 20) ? (1u << 20) : static_cast<uint32_t>(terms);
        }
    }
    ctx->crt_bits = crt_bits;
    ctx->lazy_ok = crt_bits + 7 <= (wide ? 64u : 32u);
    ctx->tight_ok = !wide && !ctx->lazy_ok && crt_bits + 4 <= 32u;
    ctx->signed_ok = !wide && crt_bits <= 24;  // signed lazy inverse butterflies: 2^6 q < 2^30 (ntt_lds.h)

    // Garner table: garner_inv[i*L + j] = (q_j)^-1 mod q_i for j < i
    // (mixed-radix CRT as the reference builds it, Runtime.cu:77-96)
    ctx->garner_inv.assign(moduli_len * moduli_len, 0);
    for (size_t i = 0; i < moduli_len; ++i)
        for (size_t j = 0; j < i; ++j)
            ctx->garner_inv[i * moduli_len + j] = h_invmod_prime(moduli[j] % moduli[i], moduli[i]);

    int rc = 0;
    e = hipMalloc(&ctx->d_limbs, sizeof(LimbConst) * moduli_len);
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_limbs, ctx->limbs.data(), sizeof(LimbConst) * moduli_len, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&ctx->d_garner, sizeof(uint64_t) * moduli_len * moduli_len);
    if (e == hipSuccess)
        e = hipMemcpy(ctx->d_garner, ctx->garner_inv.data(), sizeof(uint64_t) * moduli_len * moduli_len,
                      hipMemcpyHostToDevice);
    if (e == hipSuccess && !wide) {
        std::vector<LimbConst> lr = ctx->limbs;
        for (LimbConst &c : lr) {
            const uint64_t r = (1ull << 32) % c.q;
            c.n_inv = h_mulmod(c.n_inv, r, c.q);
            c.n_inv_sh = static_cast<uint64_t>(((u128h)c.n_inv << 32) / c.q);
            c.inv_last_w = h_mulmod(c.inv_last_w, r, c.q);
            c.inv_last_w_sh = static_cast<uint64_t>(((u128h)c.inv_last_w << 32) / c.q);
        }
        e = hipMalloc(&ctx->d_limbs_r, sizeof(LimbConst) * moduli_len);
        if (e == hipSuccess) e = hipMemcpy(ctx->d_limbs_r, lr.data(), sizeof(LimbConst) * moduli_len, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) rc = set_error(e, "context constant upload");
    if (!rc) rc = wide ? upload_tables<uint64_t>(ctx, fwd, inv) : upload_tables<uint32_t>(ctx, fwd, inv);
    if (!rc) {
        e = hipEventCreate(&ctx->timer_start);
        if (e == hipSuccess) e = hipEventCreate(&ctx->timer_stop);
        if (e != hipSuccess) rc = set_error(e, "hipEventCreate");
    }
    if (rc) {
        std::string keep = g_last_error;
        context_release(ctx);
        g_last_error = keep;
        return rc;
    }
    {
        std::lock_guard<std::mutex> reg(g_registry_mutex);
        g_contexts.push_back(ctx);
    }
    *out_ctx = ctx;
    return 0;
    ABI_GUARD_END
}

extern "C" void gpu_context_destroy(GpuContext *ctx) { context_release(ctx); }

extern "C" int gpu_context_get_N(const GpuContext *ctx, int *out_N) {
    if (!ctx || !out_N) return set_error("gpu_context_get_N: null argument");
    *out_N = ctx->N;
    return 0;
}

extern "C" int gpupoly_context_device(const GpuContext *ctx, int *out_device) {
    if (!ctx || !out_device) return set_error("gpupoly_context_device: null argument");
    *out_device = ctx->device;
    return 0;
}

// the context's compute stream (a hipStream_t), so that a host framework can order its own work - e.g. an RCCL
// collective on the engine's allocations - against the engine's on the device instead of through host syncs
extern "C" int gpupoly_context_stream(const GpuContext *ctx, void **out_stream) {
    if (!ctx || !out_stream) return set_error("gpupoly_context_stream: null argument");
    *out_stream = static_cast<void *>(ctx->stream);
    return 0;
}

// the product kernel the dispatcher chose for the last gpu_matrix_mul on this context ("" before the first one)
extern "C" const char *gpupoly_context_last_kernel(const GpuContext *ctx) { return ctx ? ctx->last_kernel.load() : ""; }

extern "C" int gpupoly_context_word_bytes(const GpuContext *ctx, int *out_bytes) {
    if (!ctx || !out_bytes) return set_error("gpupoly_context_word_bytes: null argument");
    *out_bytes = ctx->word_bytes;
    return 0;
}

// ---- events ---------------------------------------------------------------------------
extern "C" int gpu_event_set_wait(GpuEventSet *events) {
    ABI_GUARD_BEGIN
    if (!events) return 0;
    HIP_TRY(hipSetDevice(events->device));
    for (hipEvent_t ev : events->events) HIP_TRY(hipEventSynchronize(ev));
    return 0;
    ABI_GUARD_END
}

extern "C" void gpu_event_set_destroy(GpuEventSet *events) {
    if (!events) return;
    (void)hipSetDevice(events->device);
    for (hipEvent_t ev : events->events) {
        (void)hipEventSynchronize(ev);
        (void)hipEventDestroy(ev);
    }
    if (events->staging) (void)hipHostFree(events->staging);
    if (events->dev_staging && events->ctx) ctx_free(events->ctx, events->dev_staging);
    delete events;
}

// ---- device queries ---------------------------------------------------------------------
extern "C" int gpu_device_count(int *out_count) {
    ABI_GUARD_BEGIN
    if (!out_count) return set_error("gpu_device_count: null out_count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *out_count = 0;
        return set_error(e, "hipGetDeviceCount");
    }
    *out_count = n;
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_device_mem_info(int device, size_t *out_free, size_t *out_total) {
    ABI_GUARD_BEGIN
    if (!out_free || !out_total) return set_error("gpu_device_mem_info: null output");
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    HIP_TRY(hipSetDevice(device));
    hipError_t e = hipMemGetInfo(out_free, out_total);
    (void)hipSetDevice(prev);
    if (e != hipSuccess) return set_error(e, "hipMemGetInfo");
    return 0;
    ABI_GUARD_END
}

// can kernels / copies on `device` address `peer`'s memory directly (xGMI peer mapping)?  bench.py reports the matrix of a
// multi-GPU run next to its numbers; a device can always reach itself
extern "C" int gpupoly_device_can_access_peer(int device, int peer, int *out_can) {
    ABI_GUARD_BEGIN
    if (!out_can) return set_error("gpupoly_device_can_access_peer: null output");
    *out_can = 0;
    if (device == peer) {
        *out_can = 1;
        return 0;
    }
    HIP_TRY(hipDeviceCanAccessPeer(out_can, device, peer));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_device_synchronize(void) {
    ABI_GUARD_BEGIN
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    int prev = 0;
    HIP_TRY(hipGetDevice(&prev));
    for (int d = 0; d < n; ++d) {
        HIP_TRY(hipSetDevice(d));
        HIP_TRY(hipDeviceSynchronize());
    }
    HIP_TRY(hipSetDevice(prev));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpu_device_reset(void) {
    ABI_GUARD_BEGIN
    HIP_TRY(hipDeviceReset());
    return 0;
    ABI_GUARD_END
}

extern "C" void *gpu_pinned_alloc(size_t bytes) {
    if (bytes == 0) return nullptr;
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error(e, "hipHostMalloc");
        return nullptr;
    }
    return p;
}

extern "C" void gpu_pinned_free(void *ptr) {
    if (ptr) (void)hipHostFree(ptr);
}

// ---- bench timer (extension) ----------------------------------------------------------------
extern "C" int gpupoly_timer_start(GpuContext *ctx) {
    ABI_GUARD_BEGIN
    if (!ctx) return set_error("gpupoly_timer_start: null ctx");
    if (ctx_activate(ctx)) return 1;
    HIP_TRY(hipEventRecord(ctx->timer_start, ctx->stream));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpupoly_timer_stop(GpuContext *ctx, float *out_ms) {
    ABI_GUARD_BEGIN
    if (!ctx || !out_ms) return set_error("gpupoly_timer_stop: null argument");
    if (ctx_activate(ctx)) return 1;
    HIP_TRY(hipEventRecord(ctx->timer_stop, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->timer_stop));
    HIP_TRY(hipEventElapsedTime(out_ms, ctx->timer_start, ctx->timer_stop));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpupoly_timer_mark(GpuContext *ctx, uint32_t slot) {
    ABI_GUARD_BEGIN
    if (!ctx) return set_error("gpupoly_timer_mark: null ctx");
    if (slot >= 65536) return set_error("gpupoly_timer_mark: slot out of range");
    if (ctx_activate(ctx)) return 1;
    {
        std::lock_guard<std::mutex> lk(ctx->mutex);
        if (ctx->marks.size() <= slot) ctx->marks.resize(slot + 1, nullptr);
        if (!ctx->marks[slot]) HIP_TRY(hipEventCreate(&ctx->marks[slot]));
    }
    HIP_TRY(hipEventRecord(ctx->marks[slot], ctx->stream));
    return 0;
    ABI_GUARD_END
}

extern "C" int gpupoly_timer_elapsed(GpuContext *ctx, uint32_t slot_begin, uint32_t slot_end, float *out_ms) {
    ABI_GUARD_BEGIN
    if (!ctx || !out_ms) return set_error("gpupoly_timer_elapsed: null argument");
    if (slot_begin >= ctx->marks.size() || slot_end >= ctx->marks.size() || !ctx->marks[slot_begin] ||
        !ctx->marks[slot_end])
        return set_error("gpupoly_timer_elapsed: slot was never marked");
    if (ctx_activate(ctx)) return 1;
    HIP_TRY(hipEventSynchronize(ctx->marks[slot_end]));
    HIP_TRY(hipEventElapsedTime(out_ms, ctx->marks[slot_begin], ctx->marks[slot_end]));
    return 0;
    ABI_GUARD_END
}
