/*
 * libconvdr_hip.so -- C ABI of the MI355X (gfx950) kernels behind ConvDR's hot path.
 *
 * The reference (thunlp/ConvDR) is pure Python; it has no FFI of its own.  Every entry point
 * below replaces a *third-party arithmetic call site* of the reference and is
 * what a ctypes binding on the reference side would bind (INTEGRATION.md shows the stubs).
 *
 * Conventions
 *   - all pointers are DEVICE pointers (tensor.data_ptr()) unless the name says host;
 *   - plain C types only, row-major contiguous arrays, explicit sizes;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - no allocation and no synchronisation inside: the caller owns outputs and workspace;
 *   - return 0 on success, negative on error; convdr_last_error() gives the message
 *     (thread-local).  Kernels are enqueued asynchronously.
 */
#ifndef CONVDR_HIP_H
#define CONVDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* convdr_stream_t;

int convdr_version(void);
const char* convdr_last_error(void);

/* Optional per-kernel timing with hipEvents recorded on the launch stream (bench.py's roofline
 * leg).  enable(1) clears the recorded spans; collect() synchronises on the spans named `name`
 * ("ip_scan_emit", "ip_scan_sample", "ip_rescore", ...) and returns their summed duration. */
int convdr_prof_enable(int on);
int convdr_prof_collect(const char* name, float* total_ms, int* launches);

/* ------------------------------------------------------------------------------------------
 * Flat inner-product index: replaces faiss.IndexFlatIP(768) as driven by
 *   /root/reference/drivers/run_convdr_inference.py:353 (ctor), :180 (.add), :182 (.search), :202 (.reset)
 * ------------------------------------------------------------------------------------------ */

/* .add(block): build the bf16 scan copy of an fp32 block [n, d] and fold max_i ||p_i||_2 into
 * *max_norm (device float, caller zero-initialises it on reset).  d % 64 == 0.               */
int convdr_ip_prepare_block(const float* p_f32, int64_t n, int d, void* p_bf16, float* max_norm,
                            convdr_stream_t stream);

/* Bytes of device workspace convdr_ip_search needs for these sizes. */
size_t convdr_ip_workspace_bytes(int nq, int64_t n, int d, int k, int cap);

/* per-query status written by convdr_ip_search */
#define CONVDR_IP_OK 0        /* result is the exact top-k (certified)                                  */
#define CONVDR_IP_OVERFLOW 1  /* more than `cap` candidates passed tau: retry with tau_retry (raise cap
                                 when tau_retry does not exceed the tau that was used)                    */
#define CONVDR_IP_TOO_FEW 2   /* fewer than min(k, n) candidates passed tau: retry with tau_retry         */
#define CONVDR_IP_UNCERTAIN 3 /* the re-score band reaches below tau: retry with tau_retry               */

/* .search(Q, k): exact inner-product top-k of nq fp32 queries against one resident block.
 *   scan     bf16 MFMA GEMM  S~ = P_bf16 * Q_bf16^T  with a fused per-query threshold test: passages with
 *            S~ >= tau[q] are appended to a candidate list (a [nq, n] score matrix is never written);
 *            tau comes from a bf16 scan of a 1/32 sample of the block, aimed at `rank_target` hits per
 *            query (or tau_in when given);
 *   cut      with eps = 0.0079 * ||q|| * max||p|| (rigorous bound on |S~ - exact|): the band
 *            {S~ >= S~(k) - 2 eps} provably contains the exact top-k;
 *   rescore  the band is re-scored from the fp32 originals in fp64 with the canonical summation order
 *            documented in oracle/search.py, and sorted by (score desc, index asc).
 * status is OK only if the candidate list is complete over the band (cut >= tau, no overflow), so an
 * OK result equals the exhaustive exact top-k; otherwise tau_retry[q] is the threshold to pass as
 * tau_in for that query.
 * Outputs (device): D [nq, k] fp32 scores (descending), I [nq, k] int64 row indices into the block
 * (-1 / -FLT_MAX padding when n < k, as FAISS does), status [nq] int32, tau_retry [nq] fp32.
 * tau_in: NULL, or device [nq] thresholds (retry path).  cap: candidate capacity per query
 * (power of two, 1024..8192).  rank_target: expected candidates per query (0 -> 16*k, at most cap/2). */
int convdr_ip_search(const float* q_f32, int nq, const float* p_f32, const void* p_bf16, int64_t n, int d,
                     int k, const float* p_max_norm, const float* tau_in, int cap, int rank_target,
                     void* workspace, size_t workspace_bytes, float* D, int64_t* I, int32_t* status,
                     float* tau_retry, convdr_stream_t stream);

/* Instrumentation of the last convdr_ip_search on this workspace (device uint32 [nq] each):
 * candidates emitted by the scan / size of the exactly re-scored band. */
const uint32_t* convdr_ip_debug_counts(const void* workspace, int nq, int64_t n, int d, int k, int cap);
const uint32_t* convdr_ip_debug_band(const void* workspace, int nq, int64_t n, int d, int k, int cap);

#ifdef __cplusplus
}
#endif
#endif /* CONVDR_HIP_H */
