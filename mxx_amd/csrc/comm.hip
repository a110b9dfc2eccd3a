// comm.hip — the multi-GPU exchange step behind the C ABI: ONE process, N device contexts.
//
// The reference scales inside one process: a context per device (`params_for_device`, src/poly/dcrt/gpu.rs:531-557),
// rayon over the contexts (`preimage_batched_sharded`, src/sampler/trapdoor/gpu.rs:371-397), and every result that has
// to exist on another device goes through host bytes (src/lookup/ggh15/pubkey_gpu.rs:153-196).  Here the exchange is
// an all-gather of the column blocks of a column-sharded matrix, enqueued on the contexts' own streams:
//   * backend "rccl": one RCCL communicator per context (ncclCommInitAll over the contexts' devices, librccl loaded
//     on first use), ncclGroupStart / ncclAllGather per context on ITS stream / ncclGroupEnd - over xGMI;
//   * backend "peer": every context pulls its peers' blocks with device-to-device copies on its own stream, ordered
//     by events (contexts that share a device, which RCCL refuses; also selectable with MXX_HIP_COMM=peer).
// One row and equal shards: every block is a contiguous run of the full matrix, which is then the receive buffer
// itself.  Otherwise blocks are padded to the widest shard, gathered into a staging block of the context's
// stream-ordered cache and moved into place by 2-D copies.  The host never blocks; nothing here touches host memory.
#include "common.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string error;
};

// librccl is 0.5 GB of code objects: it is mapped when the first communicator is created, not when libgpupoly loads.
// A process that already holds a copy (torch bundles one under the same soname) keeps using that copy.
RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            const char *e = dlerror();
            api.error = std::string("librccl could not be loaded: ") + (e ? e : "unknown");
            return;
        }
        auto sym = [&](const char *name) {
            void *p = dlsym(api.handle, name);
            if (!p && api.error.empty()) api.error = std::string("librccl lacks ") + name;
            return p;
        };
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllGather = reinterpret_cast<decltype(api.AllGather)>(sym("ncclAllGather"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
    });
    return &api;
}

int rccl_error(RcclApi *api, ncclResult_t r, const char *what) {
    return set_error(std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
}

}  // namespace

struct GpuComm {
    std::vector<GpuContext *> ctxs;
    std::vector<ncclComm_t> comms;      // backend rccl: one per context
    std::vector<hipEvent_t> ready;      // backend peer: "this context's block is written", recorded on its stream
    std::vector<hipEvent_t> done;       //               "this context has pulled every block", recorded on its stream
    std::vector<char> peer_ok;          // [dst * n + src]: dst's kernels may read src's memory directly
    bool use_rccl = false;
    bool copy_kernel = false;           // MXX_HIP_COMM_COPY=kernel: pull with the copy kernel on one device as well
    std::mutex mutex;                   // one collective at a time per communicator
};

// rows x cols block of whole polynomials (16-byte vectors): dst / src may live on different devices (the launch runs
// on dst's device and reads src over xGMI once peer access is enabled)
__global__ void gather_rect_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t dst_pitch_vec,
                                   size_t src_pitch_vec, size_t run_vec) {
    const size_t row = blockIdx.y;
    const uint4 *s = src + row * src_pitch_vec;
    uint4 *d = dst + row * dst_pitch_vec;
    for (size_t v = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x; v < run_vec;
         v += static_cast<size_t>(gridDim.x) * blockDim.x)
        d[v] = s[v];
}

// rows runs of run_bytes from src (pitch src_pitch) to dst (pitch dst_pitch) on dst_ctx's stream
static int copy_rect(GpuComm *comm, size_t dst_rank, size_t src_rank, char *dst, size_t dst_pitch, const char *src,
                     size_t src_pitch, size_t run_bytes, size_t rows) {
    GpuContext *dctx = comm->ctxs[dst_rank], *sctx = comm->ctxs[src_rank];
    if (rows == 0 || run_bytes == 0) return 0;
    const bool same = dctx->device == sctx->device;
    const bool vec_ok = run_bytes % 16 == 0 && dst_pitch % 16 == 0 && src_pitch % 16 == 0 &&
                        reinterpret_cast<uintptr_t>(dst) % 16 == 0 && reinterpret_cast<uintptr_t>(src) % 16 == 0 && rows <= 65535;
    const bool direct = same || comm->peer_ok[dst_rank * comm->ctxs.size() + src_rank];
    if (vec_ok && direct && (!same || comm->copy_kernel)) {
        const size_t run_vec = run_bytes / 16;
        const unsigned gx = static_cast<unsigned>(std::min<size_t>((run_vec + 255) / 256, 1024));
        MXX_LAUNCH(gather_rect_kernel, dim3(gx, static_cast<unsigned>(rows)), dim3(256), 0, dctx->stream,
                           reinterpret_cast<uint4 *>(dst), reinterpret_cast<const uint4 *>(src), dst_pitch / 16,
                           src_pitch / 16, run_vec);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (same) {
        if (rows == 1 || (run_bytes == dst_pitch && run_bytes == src_pitch))
            HIP_TRY(hipMemcpyAsync(dst, src, rows * run_bytes, hipMemcpyDeviceToDevice, dctx->stream));
        else
            HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, run_bytes, rows, hipMemcpyDeviceToDevice, dctx->stream));
        return 0;
    }
    for (size_t r = 0; r < rows; ++r)  // no peer mapping: the runtime stages each run by itself
        HIP_TRY(hipMemcpyPeerAsync(dst + r * dst_pitch, dctx->device, src + r * src_pitch, sctx->device, run_bytes, dctx->stream));
    return 0;
}

extern "C" void gpupoly_comm_destroy(GpuComm *comm);

extern "C" int gpupoly_comm_create(GpuContext *const *ctxs, size_t n, GpuComm **out) {
    ABI_GUARD_BEGIN
    if (!out) return set_error("gpupoly_comm_create: null out");
    *out = nullptr;
    if (!ctxs || n == 0) return set_error("gpupoly_comm_create: no contexts");
    if (n > 64) return set_error("gpupoly_comm_create: too many contexts");
    bool distinct = true;
    for (size_t i = 0; i < n; ++i) {
        if (!ctxs[i]) return set_error("gpupoly_comm_create: null context");
        const GpuContext *a = ctxs[0], *b = ctxs[i];
        if (a->N != b->N || a->wide != b->wide || a->limb_count != b->limb_count || a->moduli != b->moduli)
            return set_error("gpupoly_comm_create: the contexts describe different rings");
        for (size_t j = 0; j < i; ++j) {
            if (ctxs[j] == ctxs[i]) return set_error("gpupoly_comm_create: duplicate context");
            if (ctxs[j]->device == ctxs[i]->device) distinct = false;
        }
    }
    const char *want = std::getenv("MXX_HIP_COMM");  // rccl | peer (default: rccl when every context has its own device)
    const bool force_peer = want && want[0] == 'p', force_rccl = want && want[0] == 'r';
    if (force_rccl && !distinct) return set_error("gpupoly_comm_create: MXX_HIP_COMM=rccl needs one device per context");
    // every failure below releases what was created so far (events, RCCL communicators) through the destroy entry point
    struct CommGuard {
        GpuComm *c;
        ~CommGuard() {
            if (c) gpupoly_comm_destroy(c);
        }
        GpuComm *operator->() const { return c; }
        GpuComm *release() {
            GpuComm *r = c;
            c = nullptr;
            return r;
        }
    } comm{new GpuComm()};
    comm->ctxs.assign(ctxs, ctxs + n);
    comm->use_rccl = distinct && !force_peer;
    if (const char *e = std::getenv("MXX_HIP_COMM_COPY")) comm->copy_kernel = e[0] == 'k';
    if (comm->use_rccl) {
        RcclApi *api = rccl_api();
        if (!api->error.empty()) return set_error("gpupoly_comm_create: " + api->error);
        std::vector<int> devs(n);
        for (size_t i = 0; i < n; ++i) devs[i] = ctxs[i]->device;
        comm->comms.assign(n, nullptr);
        ncclResult_t r = api->CommInitAll(comm->comms.data(), static_cast<int>(n), devs.data());
        if (r != ncclSuccess) return rccl_error(api, r, "ncclCommInitAll");
    } else {
        comm->ready.assign(n, nullptr);
        comm->done.assign(n, nullptr);
        comm->peer_ok.assign(n * n, 0);
        for (size_t i = 0; i < n; ++i) {
            HIP_TRY(hipSetDevice(ctxs[i]->device));
            HIP_TRY(hipEventCreateWithFlags(&comm->ready[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&comm->done[i], hipEventDisableTiming));
            for (size_t j = 0; j < n; ++j) {
                if (ctxs[j]->device == ctxs[i]->device) {
                    comm->peer_ok[i * n + j] = 1;
                    continue;
                }
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, ctxs[i]->device, ctxs[j]->device));
                if (!can) continue;
                hipError_t e = hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) {
                    (void)hipGetLastError();
                    e = hipSuccess;
                }
                if (e != hipSuccess) return set_error(e, "hipDeviceEnablePeerAccess");
                comm->peer_ok[i * n + j] = 1;
            }
        }
    }
    *out = comm.release();
    return 0;
    ABI_GUARD_END
}

extern "C" void gpupoly_comm_destroy(GpuComm *comm) {
    if (!comm) return;
    if (comm->use_rccl) {
        RcclApi *api = rccl_api();
        for (size_t i = 0; i < comm->comms.size(); ++i) {
            if (!comm->comms[i]) continue;
            // a communicator must not go away under its own collective - but its contexts may be gone already (a host
            // wrapper finalised at interpreter shutdown): the raw pointers are only followed while still registered
            if (ctx_is_registered(comm->ctxs[i])) {
                (void)hipSetDevice(comm->ctxs[i]->device);
                (void)hipStreamSynchronize(comm->ctxs[i]->stream);
            }
            (void)api->CommDestroy(comm->comms[i]);
        }
    }
    for (size_t i = 0; i < comm->ready.size(); ++i) {
        if (ctx_is_registered(comm->ctxs[i])) (void)hipSetDevice(comm->ctxs[i]->device);
        if (comm->ready[i]) (void)hipEventDestroy(comm->ready[i]);
        if (comm->done[i]) (void)hipEventDestroy(comm->done[i]);
    }
    delete comm;
}

extern "C" int gpupoly_comm_size(const GpuComm *comm, int *out_size) {
    if (!comm || !out_size) return set_error("gpupoly_comm_size: null argument");
    *out_size = static_cast<int>(comm->ctxs.size());
    return 0;
}

extern "C" const char *gpupoly_comm_backend(const GpuComm *comm) { return !comm ? "" : (comm->use_rccl ? "rccl" : "peer"); }

extern "C" int gpupoly_matrix_all_gather_columns(GpuComm *comm, const GpuMatrix *const *local_blocks, GpuMatrix *const *full) {
    ABI_GUARD_BEGIN
    if (!comm || !local_blocks || !full) return set_error("gpupoly_matrix_all_gather_columns: null argument");
    const size_t n = comm->ctxs.size();
    size_t cols_total = 0, max_cols = 0;
    std::vector<size_t> col_start(n);
    for (size_t r = 0; r < n; ++r) {
        const GpuMatrix *b = local_blocks[r];
        const GpuMatrix *f = full[r];
        if (!b || !f) return set_error("gpupoly_matrix_all_gather_columns: null matrix");
        if (b->ctx != comm->ctxs[r] || f->ctx != comm->ctxs[r])
            return set_error("gpupoly_matrix_all_gather_columns: matrix r must live in context r of the communicator");
        if (b->level != local_blocks[0]->level || f->level != b->level)
            return set_error("gpupoly_matrix_all_gather_columns: level mismatch");
        if (b->rows != local_blocks[0]->rows || f->rows != b->rows)
            return set_error("gpupoly_matrix_all_gather_columns: row count mismatch");
        if (b->format != local_blocks[0]->format)
            return set_error("gpupoly_matrix_all_gather_columns: the blocks are in different formats");
        if (b == f) return set_error("gpupoly_matrix_all_gather_columns: a block must not alias its output");
        col_start[r] = cols_total;
        cols_total += b->cols;
        max_cols = std::max(max_cols, b->cols);
    }
    for (size_t r = 0; r < n; ++r)
        if (full[r]->cols != cols_total)
            return set_error("gpupoly_matrix_all_gather_columns: output must have the sum of the blocks' columns");
    const int fmt = local_blocks[0]->format;
    const size_t rows = local_blocks[0]->rows;
    const size_t poly_bytes = matrix_limbs(local_blocks[0]) * static_cast<size_t>(comm->ctxs[0]->N) * comm->ctxs[0]->word_bytes;
    std::lock_guard<std::mutex> lk(comm->mutex);
    // RCCL moves bytes: the tag travels here - once everything is enqueued, so an error return leaves the outputs' tags
    // alone, and under the communicator's mutex
    auto retag = [&]() {
        for (size_t r = 0; r < n; ++r) full[r]->format = fmt;
        return 0;
    };
    if (rows == 0 || cols_total == 0) return retag();

    if (!comm->use_rccl) {
        // pull: context r copies every block into its own full matrix on its own stream, after the event that marks the
        // block as written; afterwards every source waits for its readers, so a block may be overwritten or freed right
        // after this call (stream order does the rest)
        for (size_t p = 0; p < n; ++p) {
            HIP_TRY(hipSetDevice(comm->ctxs[p]->device));
            HIP_TRY(hipEventRecord(comm->ready[p], comm->ctxs[p]->stream));
        }
        for (size_t r = 0; r < n; ++r) {
            GpuContext *ctx = comm->ctxs[r];
            HIP_TRY(hipSetDevice(ctx->device));
            for (size_t q = 0; q < n; ++q) {
                const size_t p = (r + q) % n;  // own block first, then the peers in a staggered order
                const GpuMatrix *b = local_blocks[p];
                if (b->cols == 0) continue;
                if (p != r) HIP_TRY(hipStreamWaitEvent(ctx->stream, comm->ready[p], 0));
                char *dst = static_cast<char *>(full[r]->data) + col_start[p] * poly_bytes;
                if (copy_rect(comm, r, p, dst, cols_total * poly_bytes, static_cast<const char *>(b->data), b->cols * poly_bytes,
                              b->cols * poly_bytes, rows))
                    return 1;
            }
            HIP_TRY(hipEventRecord(comm->done[r], ctx->stream));
        }
        for (size_t p = 0; p < n; ++p) {
            HIP_TRY(hipSetDevice(comm->ctxs[p]->device));
            for (size_t r = 0; r < n; ++r)
                if (r != p) HIP_TRY(hipStreamWaitEvent(comm->ctxs[p]->stream, comm->done[r], 0));
        }
        return retag();
    }

    RcclApi *api = rccl_api();
    bool equal = true;
    for (size_t r = 0; r < n; ++r) equal = equal && local_blocks[r]->cols == max_cols;
    const bool direct = rows == 1 && equal;
    const size_t slot_bytes = rows * max_cols * poly_bytes;  // one rank's padded block
    std::vector<std::unique_ptr<CtxBlock>> send(n), recv(n);
    std::vector<const void *> sendp(n);
    std::vector<void *> recvp(n);
    for (size_t r = 0; r < n; ++r) {
        GpuContext *ctx = comm->ctxs[r];
        HIP_TRY(hipSetDevice(ctx->device));
        const GpuMatrix *b = local_blocks[r];
        if (direct) {
            sendp[r] = b->data;
            recvp[r] = full[r]->data;
            continue;
        }
        recv[r].reset(new CtxBlock(ctx));
        if (recv[r]->alloc(n * slot_bytes)) return 1;
        recvp[r] = recv[r]->ptr;
        if (b->cols == max_cols) {
            sendp[r] = b->data;  // a full-width block is its own send buffer
        } else {
            send[r].reset(new CtxBlock(ctx));
            if (send[r]->alloc(slot_bytes)) return 1;
            sendp[r] = send[r]->ptr;
            if (copy_rect(comm, r, r, static_cast<char *>(send[r]->ptr), max_cols * poly_bytes, static_cast<const char *>(b->data),
                          b->cols * poly_bytes, b->cols * poly_bytes, rows))
                return 1;
        }
    }
    ncclResult_t res = api->GroupStart();
    if (res != ncclSuccess) return rccl_error(api, res, "ncclGroupStart");
    for (size_t r = 0; r < n && res == ncclSuccess; ++r)
        res = api->AllGather(sendp[r], recvp[r], slot_bytes, ncclUint8, comm->comms[r], comm->ctxs[r]->stream);
    const ncclResult_t end = api->GroupEnd();
    if (res != ncclSuccess) return rccl_error(api, res, "ncclAllGather");
    if (end != ncclSuccess) return rccl_error(api, end, "ncclGroupEnd");
    if (!direct) {
        for (size_t r = 0; r < n; ++r) {
            HIP_TRY(hipSetDevice(comm->ctxs[r]->device));
            for (size_t p = 0; p < n; ++p) {
                const size_t c = local_blocks[p]->cols;
                if (c == 0) continue;
                char *dst = static_cast<char *>(full[r]->data) + col_start[p] * poly_bytes;
                const char *src = static_cast<const char *>(recvp[r]) + p * slot_bytes;
                if (copy_rect(comm, r, r, dst, cols_total * poly_bytes, src, max_cols * poly_bytes, c * poly_bytes, rows)) return 1;
            }
        }
    }
    return retag();  // the staging blocks return to their contexts' caches here, stream-ordered behind their readers
    ABI_GUARD_END
}
