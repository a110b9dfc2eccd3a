/*
 * gpupoly.h — C ABI of libgpupoly for AMD MI355X (gfx950).
 *
 * Drop-in boundary for the `gpu` feature of MachinaIO/mxx: every entry point
 * below replaces the function of the same name that mxx's Rust side binds in
 *   src/poly/dcrt/gpu.rs:69-240  (unsafe extern "C" block)
 * and that its CUDA tree declares in
 *   cuda/include/Runtime.cuh:20-44,108
 *   cuda/include/matrix/MatrixData.cuh:10-27
 *   cuda/include/matrix/MatrixArith.cuh:10-26
 *   cuda/include/matrix/MatrixNTT.cuh:9-10
 *   cuda/include/matrix/MatrixDecompose.cuh:26-37
 *   cuda/include/matrix/MatrixSampling.cuh:25-37
 *   cuda/include/matrix/MatrixTrapdoor.cuh:66-101
 *   cuda/include/matrix/MatrixSerde.cuh:10-58
 * (paths relative to the reference checkout).
 *
 * Conventions:
 *   - int-returning functions: 0 = ok, non-zero = error; the message is
 *     available from gpu_last_error() (thread-local, valid until the next
 *     error on that thread).  Nothing throws across this boundary.
 *   - *_create hands out an owning pointer; *_destroy is stream-ordered and
 *     may be called while device work on the object is still in flight.
 *   - handles may be used concurrently from several host threads.
 *   - compute entry points do not block the host; load/store *_batch return
 *     an event set (possibly NULL) the caller waits on and destroys;
 *     gpu_matrix_equal and the compact-bytes pair are synchronous.
 *   - device layout is private: words [poly][limb][N], poly = row*cols + col,
 *     uint32_t residues when every modulus is < 2^31, uint64_t otherwise.
 *   - EVAL format is OpenFHE's: slot k of limb i holds a(psi_i^(2*bitrev(k)+1))
 *     with psi_i the minimum primitive 2N-th root mod q_i (the reference CPU
 *     path's convention), so EVAL bytes are interchangeable with the CPU side.
 */
#ifndef GPUPOLY_H
#define GPUPOLY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct GpuContext GpuContext;
typedef struct GpuMatrix GpuMatrix;
typedef struct GpuEventSet GpuEventSet;
typedef struct GpuP1CovarianceCache GpuP1CovarianceCache;
typedef struct GpuComm GpuComm; /* extension: communicator over the device contexts of ONE process */

/* src/poly/dcrt/gpu.rs:45-61, cuda/include/ChaCha.cuh:9-12 — passed BY VALUE */
typedef struct GpuRngSeed {
    uint64_t words[4];
} GpuRngSeed;

/* src/poly/dcrt/gpu.rs:242-247 */
#define GPU_POLY_FORMAT_COEFF 0
#define GPU_POLY_FORMAT_EVAL 1
#define GPU_MATRIX_DIST_UNIFORM 0
#define GPU_MATRIX_DIST_GAUSS 1
#define GPU_MATRIX_DIST_BIT 2
#define GPU_MATRIX_DIST_TERNARY 3

/* ---- runtime: cuda/include/Runtime.cuh:20-44,108 ------------------------- */
/* L = moduli_len - 1 (top level); gpu_ids[0] is the device the context lives on. */
int gpu_context_create(uint32_t logN, uint32_t L, uint32_t dnum, const uint64_t *moduli, size_t moduli_len,
                       const int *gpu_ids, size_t gpu_ids_len, GpuContext **out_ctx);
void gpu_context_destroy(GpuContext *ctx);
int gpu_context_get_N(const GpuContext *ctx, int *out_N);

int gpu_event_set_wait(GpuEventSet *events);
void gpu_event_set_destroy(GpuEventSet *events);

int gpu_device_count(int *out_count);
int gpu_device_mem_info(int device, size_t *out_free, size_t *out_total);
int gpu_device_synchronize(void);
int gpu_device_reset(void);

const char *gpu_last_error(void);
int gpu_set_last_error(const char *msg);

void *gpu_pinned_alloc(size_t bytes);
void gpu_pinned_free(void *ptr);

/* ---- storage: cuda/include/matrix/MatrixData.cuh:10-27 -------------------- */
/* contents undefined after create; a matrix at `level` uses limbs 0..=level */
int gpu_matrix_create(GpuContext *ctx, int level, size_t rows, size_t cols, int format, GpuMatrix **out);
void gpu_matrix_destroy(GpuMatrix *mat);
int gpu_matrix_copy(GpuMatrix *dst, const GpuMatrix *src);
int gpu_matrix_copy_block(GpuMatrix *out, const GpuMatrix *src, size_t dst_row, size_t dst_col, size_t src_row,
                          size_t src_col, size_t rows, size_t cols);

/* ---- arithmetic: cuda/include/matrix/MatrixArith.cuh:10-26 ---------------- */
int gpu_matrix_add(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);
int gpu_matrix_sub(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);
int gpu_matrix_add_block(GpuMatrix *out, const GpuMatrix *src, size_t dst_row, size_t dst_col, size_t src_row,
                         size_t src_col, size_t rows, size_t cols);
int gpu_matrix_mul(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);          /* both EVAL */
int gpu_matrix_mul_scalar(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *scalar); /* scalar is 1x1, EVAL */
int gpu_matrix_equal(const GpuMatrix *lhs, const GpuMatrix *rhs, int *out_equal);

/* ---- NTT: cuda/include/matrix/MatrixNTT.cuh:9-10 (idempotent) -------------- */
int gpu_matrix_ntt_all(GpuMatrix *mat);
int gpu_matrix_intt_all(GpuMatrix *mat);

/* ---- gadget / decompose: cuda/include/matrix/MatrixDecompose.cuh:26-37 ---- */
int gpu_matrix_fill_gadget(GpuMatrix *out, uint32_t base_bits);
int gpu_matrix_fill_small_gadget(GpuMatrix *out, uint32_t base_bits);
int gpu_matrix_fill_small_decomposed_identity_chunk(GpuMatrix *out, const GpuMatrix *scalar_by_digit,
                                                    size_t chunk_idx);
int gpu_matrix_decompose_base(const GpuMatrix *src, uint32_t base_bits, GpuMatrix *out);
int gpu_matrix_decompose_base_small(const GpuMatrix *src, uint32_t base_bits, GpuMatrix *out);

/* ---- sampling: MatrixSampling.cuh:25-37, MatrixTrapdoor.cuh:66-101 --------- */
int gpu_matrix_sample_distribution(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed);
int gpu_matrix_sample_distribution_columns(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed,
                                           size_t full_ncol, size_t col_offset);
int gpu_matrix_gauss_samp_gq_arb_base(GpuMatrix *src, uint32_t base_bits, double c, double dgg_stddev,
                                      GpuRngSeed seed, GpuMatrix *out);
int gpu_matrix_sample_p1_full(const GpuMatrix *a_mat, const GpuMatrix *b_mat, const GpuMatrix *d_mat,
                              const GpuMatrix *tp2, double sigma, double s, double dgg_stddev, GpuRngSeed seed,
                              GpuMatrix *out);
int gpu_matrix_create_p1_covariance_cache(const GpuMatrix *a_mat, const GpuMatrix *b_mat, const GpuMatrix *d_mat,
                                          double sigma, double s, double dgg_stddev,
                                          GpuP1CovarianceCache **out_cache);
void gpu_matrix_destroy_p1_covariance_cache(GpuP1CovarianceCache *cache);
int gpu_matrix_sample_p1_full_cached(const GpuP1CovarianceCache *cache, const GpuMatrix *tp2, GpuRngSeed seed,
                                     GpuMatrix *out);

/* ---- serde: cuda/include/matrix/MatrixSerde.cuh:10-58 ---------------------- */
/* host layout [poly][limb][N] little-endian u64, poly stride = bytes_per_poly */
int gpu_matrix_load_rns_batch(GpuMatrix *mat, const uint8_t *bytes, size_t bytes_per_poly, int format,
                              GpuEventSet **out_events);
int gpu_matrix_store_rns_batch(const GpuMatrix *mat, uint8_t *bytes_out, size_t bytes_per_poly, int format,
                               GpuEventSet **out_events);
int gpu_matrix_store_const_coeff_batch(const GpuMatrix *mat, uint64_t *words_out, size_t words_per_poly,
                                       GpuEventSet **out_events);
/* compact wire format (coefficient-domain, CRT-reconstructed, centred, bit-packed at the matrix-wide width): the store
 * takes an EVAL matrix to the coefficient domain in place.  When payload_capacity is too small the call fails with
 * "payload buffer too small ..." AND reports the width / length it needs through the three out parameters (the
 * reference reports only the error), so a host need not reserve the worst case of bits(Q) per coefficient.       */
int gpu_matrix_store_compact_bytes(GpuMatrix *mat, uint8_t *payload_out, size_t payload_capacity,
                                   uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                   size_t *out_payload_len);
int gpu_matrix_load_compact_bytes(GpuMatrix *mat, const uint8_t *payload, size_t payload_len,
                                  uint16_t max_coeff_bits);
int gpu_poly_store_compact_bytes(GpuMatrix *poly, uint8_t *payload_out, size_t payload_capacity,
                                 uint16_t *out_max_coeff_bits, uint16_t *out_bytes_per_coeff,
                                 size_t *out_payload_len);
int gpu_poly_load_compact_bytes(GpuMatrix *poly, const uint8_t *payload, size_t payload_len,
                                uint16_t max_coeff_bits);

/* ---- MI355X extensions (not in the reference ABI; prefixed gpupoly_) ------- */
/* S * G^-1(B) in one call (replaces the Rust-side loop src/matrix/gpu_dcrt_poly.rs:1414-1493, which re-reads S for
 * every column chunk): digits are generated inside the forward transform's load for all columns at once when
 * memory allows, then one product into `out`.  The EVAL-form digit matrix IS written once and read once (a full
 * fusion would need 8 x 16384 accumulators per workgroup).                          */
int gpupoly_matrix_mul_decompose(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs, uint32_t base_bits);
/* outs[i] = lhss[i] * rhss[i], i < count: independent products of one context and level (all EVAL) in one call.  Small
 * products - a level of circuit gates on a small ring, where every product is a launch-latency-bound kernel - go out
 * up to 64 per launch; large ones run one by one through the tuned kernels.  No output may be another product's
 * operand.  (the reference issues one call per gate, src/circuit/poly_circuit/eval.rs:269.)        */
int gpupoly_matrix_mul_batch(GpuMatrix *const *outs, const GpuMatrix *const *lhss, const GpuMatrix *const *rhss, size_t count);
/* A level of independent circuit gates in one call (the reference issues one ABI call per gate,
 * src/circuit/poly_circuit/eval.rs:269-345): products go out up to 64 per launch (gpupoly_matrix_mul_batch), the
 * point-wise gates - add, sub, negate, product by a 1x1 ring element (Small / LargeScalarMul) - up to 64 per launch,
 * decompositions through their tuned paths.  One context; no output may be another gate's operand or output; formats
 * and shapes as for the single-gate entry points named below.                                                      */
#define GPUPOLY_OP_MUL 0           /* gpu_matrix_mul(out, lhs, rhs) */
#define GPUPOLY_OP_ADD 1           /* gpu_matrix_add(out, lhs, rhs); out may be lhs */
#define GPUPOLY_OP_SUB 2           /* gpu_matrix_sub */
#define GPUPOLY_OP_MUL_SCALAR 3    /* gpu_matrix_mul_scalar(out, lhs, rhs = 1x1) */
#define GPUPOLY_OP_NEG 4           /* gpupoly_matrix_neg(out, lhs); rhs ignored */
#define GPUPOLY_OP_DECOMPOSE 5     /* gpu_matrix_decompose_base(lhs, base_bits, out); rhs ignored */
#define GPUPOLY_OP_MUL_DECOMPOSE 6 /* gpupoly_matrix_mul_decompose(out, lhs, rhs, base_bits) */
typedef struct GpuBatchOp {
    int kind;
    GpuMatrix *out;
    const GpuMatrix *lhs;
    const GpuMatrix *rhs;
} GpuBatchOp;
int gpupoly_batch(const GpuBatchOp *ops, size_t count, uint32_t base_bits);
/* lhs * small-G^-1(rhs) (digits of limb 0 only): replaces the column-chunk loop of mul_decompose_small
 * (src/matrix/gpu_dcrt_poly.rs:1495-1574).                                                                */
int gpupoly_matrix_mul_decompose_small(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs, uint32_t base_bits);
/* lhs * (I_identity_size (x) rhs): one product per identity block, written in place, no slice / concat copies for
 * row-vector operands (replaces src/matrix/gpu_dcrt_poly.rs:1374-1390).                                     */
int gpupoly_matrix_mul_tensor_identity(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs, size_t identity_size);
/* lhs * (I_identity_size (x) G^-1(rhs)): G^-1(rhs) is built ONCE and reused by every identity block; the reference
 * decomposes every column again for every block (src/matrix/gpu_dcrt_poly.rs:1392-1412).                    */
int gpupoly_matrix_mul_tensor_identity_decompose(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs,
                                                 size_t identity_size, uint32_t base_bits);
/* out <- INTT(lhs o scalar_1x1): the point-wise product rides in the inverse transform's load (one HBM round
 * trip instead of two; replaces gpu_matrix_mul_scalar + gpu_matrix_intt_all).  out may be lhs.           */
int gpupoly_matrix_mul_scalar_intt(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *scalar_1x1);
/* out = src^T in one launch (the reference's wrapper issues rows*cols single-polynomial copy_block calls,
 * src/matrix/gpu_dcrt_poly.rs:1190-1199).                                                                */
int gpupoly_matrix_transpose(GpuMatrix *out, const GpuMatrix *src);
/* Constants written on the device instead of uploaded as full-size host byte vectors
 * (src/matrix/gpu_dcrt_poly.rs:343-365 `new_zero_with_state`, :1158-1188 `identity`).  fill_zero keeps the format
 * tag; fill_identity puts scalar_1x1 (EVAL; NULL = the constant 1) on the diagonal and tags the result EVAL.  */
/* out = lhs (x) rhs (Kronecker product, all EVAL) in one launch; the reference's wrapper runs an entry slice, a
 * mul_scalar and a copy_block per entry of lhs (src/matrix/gpu_dcrt_poly.rs:1225-1252).                     */
int gpupoly_matrix_tensor(GpuMatrix *out, const GpuMatrix *lhs, const GpuMatrix *rhs);
/* out[dst_row .. dst_row + lhs.rows) = lhs + rhs: the sum lands in a row block of a taller matrix (contiguous in
 * the row-major layout) instead of a copy_block followed by an add_block; used by the preimage's final assembly
 * (src/sampler/trapdoor/gpu.rs:340-369).  The whole destination takes the operands' format tag.            */
int gpupoly_matrix_add_rows(GpuMatrix *out, size_t dst_row, const GpuMatrix *lhs, const GpuMatrix *rhs);
/* out[dst_row .. dst_row + coeff.rows) = NTT(coeff) + addend: `coeff` is a COEFF matrix, `addend` an EVAL matrix of the
 * same shape; transform, sum and placement are one pass at n = 2^14 with 32-bit words.  Elsewhere, consume_coeff != 0
 * (the caller gives `coeff` up: contents and tag unspecified afterwards) lets the library transform it in place and add
 * in a second pass; with consume_coeff == 0 `coeff` is left untouched at the price of a copy.  The preimage's bottom
 * block p2 + z (src/sampler/trapdoor/gpu.rs:340-369).  The whole destination is tagged EVAL.               */
int gpupoly_matrix_ntt_add_rows(GpuMatrix *out, size_t dst_row, GpuMatrix *coeff, const GpuMatrix *addend,
                                int consume_coeff);
/* A matrix object over rows [row, row + rows) of m (contiguous in the row-major layout) that SHARES m's storage - an
 * operand without the copy a slice makes.  Destroy it with gpu_matrix_destroy (the storage stays m's) before m.  */
int gpupoly_matrix_row_view(GpuMatrix *m, size_t row, size_t rows, GpuMatrix **out_view);
/* out = -src in one pass (the reference's wrapper: upload zeros, clone, subtract - gpu_dcrt_poly.rs:1890-1897). */
int gpupoly_matrix_neg(GpuMatrix *out, const GpuMatrix *src);
int gpupoly_matrix_fill_zero(GpuMatrix *out);
int gpupoly_matrix_fill_identity(GpuMatrix *out, const GpuMatrix *scalar_1x1);
/* G^-1 of a freshly sampled rows x cols matrix: out is (rows*k) x cols, k = digits per entry (small != 0: the digits
 * of limb 0 only).  Same samples as gpu_matrix_sample_distribution, same digits as gpu_matrix_decompose_base(_small);
 * the sample's NTT and the decomposition's copy + INTT are skipped (replaces the pairs in
 * src/sampler/gpu.rs:91-115, `sample_hash_decomposed` / `sample_hash_small_decomposed`).                   */
int gpupoly_matrix_sample_decomposed(GpuMatrix *out, int dist_type, double sigma, GpuRngSeed seed, uint32_t base_bits,
                                     int small);
/* hipEvent timing on the context's compute stream (bench.py's roofline leg). */
int gpupoly_timer_start(GpuContext *ctx);
int gpupoly_timer_stop(GpuContext *ctx, float *out_ms);
/* Non-blocking event marks on the compute stream (slot < 65536) and the elapsed
 * time between two marks once both have completed (blocks on the later one).    */
int gpupoly_timer_mark(GpuContext *ctx, uint32_t slot);
int gpupoly_timer_elapsed(GpuContext *ctx, uint32_t slot_begin, uint32_t slot_end, float *out_ms);
/* raw device pointer / byte size of a matrix (zero-copy interop, e.g. RCCL).  */
int gpupoly_matrix_device_ptr(const GpuMatrix *mat, void **out_ptr, size_t *out_bytes);
/* Replica of `src` in another context (same ring; any device): one device-to-device / peer copy over
 * xGMI, ordered on both contexts' streams - replaces the reference's host round trip
 * to_cpu_staging_bytes -> from_cpu_staging_bytes (src/lookup/ggh15/pubkey_gpu.rs:153-196).     */
int gpupoly_matrix_copy_to_context(GpuContext *dst_ctx, const GpuMatrix *src, GpuMatrix **out);
int gpupoly_context_device(const GpuContext *ctx, int *out_device);
int gpupoly_context_word_bytes(const GpuContext *ctx, int *out_bytes);
/* name of the product kernel the dispatcher launched for the last gpu_matrix_mul on this context (bench.py labels its
 * roofline with what actually ran); "" before the first product                                                  */
const char *gpupoly_context_last_kernel(const GpuContext *ctx);
/* the context's compute stream (hipStream_t): lets the host order collectives against engine work on the device */
int gpupoly_context_stream(const GpuContext *ctx, void **out_stream);
/* ---- multi-GPU exchange, one process / N device contexts ----------------------------------------
 * The reference runs ONE process with a context per device (`params_for_device`, src/poly/dcrt/gpu.rs:531-557) and
 * rayon over them (`preimage_batched_sharded`, src/sampler/trapdoor/gpu.rs:371-397); whatever has to exist on another
 * device travels through host bytes.  A communicator binds such contexts (the same ring on every one).  Backend
 * "rccl": ncclCommInitAll over the contexts' devices (librccl is loaded when the first communicator is created),
 * collectives enqueued on each context's own stream, xGMI between the devices.  Backend "peer": device-to-device
 * pulls ordered by events - chosen when contexts share a device (RCCL refuses that) or with MXX_HIP_COMM=peer.     */
int gpupoly_comm_create(GpuContext *const *ctxs, size_t n, GpuComm **out_comm);
void gpupoly_comm_destroy(GpuComm *comm);
int gpupoly_comm_size(const GpuComm *comm, int *out_size);
const char *gpupoly_comm_backend(const GpuComm *comm); /* "rccl" or "peer" */
/* All-gather of the column blocks of a column-sharded matrix: local_blocks[r] (rows x c_r, in context r, any format,
 * the same one everywhere) lands in columns [c_0 + .. + c_(r-1), +c_r) of full[s] (rows x sum c_r, in context s) for
 * every s; full[s] takes the blocks' format tag.  Shards may be uneven or empty.  Enqueued on the contexts' streams
 * behind whatever produced the blocks; the host does not block; a block may be overwritten or destroyed right after
 * the call.  One row and equal shards gather straight into full[s]; other shapes go through a padded staging block
 * of the context's allocator.  One collective at a time per communicator (calls serialise on its mutex); other host
 * threads may keep enqueueing work on the contexts' streams meanwhile, as long as nothing enqueued after the call
 * writes a block that was passed to it before the call returns.  The outputs' format tags change only on success. */
int gpupoly_matrix_all_gather_columns(GpuComm *comm, const GpuMatrix *const *local_blocks, GpuMatrix *const *full);
/* ---- several independent requests in one launch (replaces the rayon fan-out of small requests,
 * src/sampler/trapdoor/gpu.rs:371-397, whose callers - src/lookup/ggh15/pubkey_gpu.rs:615-971 - hand it dozens of
 * few-column targets against one trapdoor).  A matrix is read as the column-wise concatenation of `nseg` segments
 * (1..64), segment j = the next seg_cols[j] columns, with its own seed seeds[j]; the columns of segment j come out
 * bit for bit as the plain entry point writes them for a matrix made of those columns alone under seeds[j] (every
 * element's stream is keyed by its position inside its segment).  Gaussian distribution / trapdoor dimension <= 2 /
 * at most four digits per tower / n a multiple of 64 (128 for _sample_distribution_segments); anything else returns
 * an error whose text contains "unsupported" and the caller issues the requests one by one.                      */
int gpupoly_matrix_sample_distribution_segments(GpuMatrix *out, int dist_type, double sigma, const GpuRngSeed *seeds,
                                                const size_t *seg_cols, size_t nseg);
int gpupoly_matrix_sample_p1_full_cached_segments(const GpuP1CovarianceCache *cache, const GpuMatrix *tp2,
                                                  const GpuRngSeed *seeds, const size_t *seg_cols, size_t nseg,
                                                  GpuMatrix *out);
int gpupoly_matrix_gauss_samp_gq_arb_base_segments(GpuMatrix *src, uint32_t base_bits, double c, double dgg_stddev,
                                                   const GpuRngSeed *seeds, const size_t *seg_cols, size_t nseg,
                                                   GpuMatrix *out);
/* out = [blocks[0] | blocks[1] | ...] / blocks[j] = the next blocks[j]->cols columns of src, in one launch per 64
 * blocks (the wrapper's concat_columns / slice_columns are a gpu_matrix_copy_block launch per block,
 * src/matrix/gpu_dcrt_poly.rs:1216-1260).  Same rows, level and context everywhere; the written side takes the read
 * side's format tag.                                                                                              */
int gpupoly_matrix_concat_columns(GpuMatrix *out, const GpuMatrix *const *blocks, size_t n);
int gpupoly_matrix_split_columns(const GpuMatrix *src, GpuMatrix *const *blocks, size_t n);
/* kernel launches issued by the library since it was loaded (every context; copies / memsets not counted): bench.py
 * reports launches per step for the launch-bound small-ring chain                                              */
uint64_t gpupoly_launch_count(void);
/* Test instrument: evaluates the samplers' deterministic math (mxx_amd/csrc/detmath.h) ON THE DEVICE for n host-supplied
 * doubles - fn 0: log(x), 1: cos(2 pi x), 2: sqrt(-2 log x) - and copies the results back (synchronous).  The Box-Muller
 * step of the G-lattice sampler (cuda/src/matrix/MatrixTrapdoor.cu:701-833 calls log / cos there) is the only place on the
 * path with transcendental functions; this lets a test hold the device's results to libm (log and sqrt(-2 log) within 2 ulp, cos(2 pi x) within 3). */
int gpupoly_detmath_eval(GpuContext *ctx, int fn, const double *host_in, double *host_out, size_t n);
/* 1 if `device` can address `peer`'s memory directly (xGMI peer mapping; a device always reaches itself) */
int gpupoly_device_can_access_peer(int device, int peer, int *out_can);
/* a one-thread no-op kernel (`gpupoly_marker_kernel`) on the context's stream: delimits bench.py's timed region in a
 * profiler's dispatch list (tools/pmc_window.py counts only what lies between two markers)                     */
int gpupoly_marker_launch(GpuContext *ctx, uint32_t id);
/* Launch trace (bench.py's composed roofline of a multi-kernel call: a preimage, a chain step).  Between _begin and
 * _end every kernel launch and every device-to-device copy of the library - any context, any host thread - is
 * bracketed by two hipEvents on the stream it is enqueued on.  _end stops recording, waits for the recorded work and
 * returns one line per launch in launch order: "kernel \t blocks \t threads \t algorithmic bytes \t ms \n" (bytes = the
 * launch's operands read once + written once, 0 where the launcher does not state them).  The string belongs to the
 * library and stays valid until the next _begin / _end; NULL on error.  Tracing costs two event records per launch:
 * durations are the kernels' own, the call's wall time is not what an untraced call takes.                       */
int gpupoly_trace_begin(void);
const char *gpupoly_trace_end(void);
const char *gpupoly_version(void);
/* MXX_HIP_* switches are read once, at gpu_context_create; this re-reads them for every live
 * context of the process (tests flip them between calls).                      */
int gpupoly_reload_env(void);

#ifdef __cplusplus
}
#endif
#endif /* GPUPOLY_H */
