/*
 * spml_hip.h -- C-ABI of libspml_hip.so: the MI355X (gfx950) kernels behind the
 * SPML pixel-to-segment contrastive hot path.
 *
 * The reference (twke18/SPML) has no FFI: its hot path is a set of Python
 * callables made of ATen op chains.  Each entry point
 * below replaces one such chain; the reference file:line it replaces is cited
 * on the declaration.  The Python mirrors under spml_amd/ bind these with
 * ctypes (spml_amd/_ffi.py); INTEGRATION.md shows the stub a reference
 * maintainer would add.
 *
 * Conventions (every function):
 *   - plain device pointers + explicit sizes; row-major, densely packed;
 *   - float data is fp32, labels/indices are int64 at the boundary (the
 *     reference API is int64 everywhere), int32 inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *     launches are stream-ordered, nothing synchronises, nothing allocates;
 *   - scratch comes from the caller: `ws`/`ws_bytes`, sized by the matching
 *     `*_workspace_bytes()` query (host-only, no device access);
 *   - return value: SPML_OK (0) or a negative spml_status; never throws.
 */
#ifndef SPML_HIP_H_
#define SPML_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum spml_status {
  SPML_OK = 0,
  SPML_ERR_INVALID_ARG = -1,   /* null pointer, negative size, bad alignment */
  SPML_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels cover        */
  SPML_ERR_WORKSPACE = -3,     /* workspace missing or too small              */
  SPML_ERR_LAUNCH = -4         /* hipGetLastError() != hipSuccess after launch */
} spml_status;

/* Human-readable text for a status code (static storage). */
const char* spml_status_string(int status);

/* ABI version: bumped on any signature change. */
/* Bumped whenever an entry point changes its arguments or a flag its meaning; spml_amd/_ffi.py refuses a library
 * whose version differs from the header it was written against.  2: round 4 (count_dev in the batch-norm backward,
 * spml_bn_finalize_ranks_f32); 3: round 5 (SPML_KMEANS_NO_PASS64 / _TWO_KERNEL_FINALIZE / _NO_V4K, paths "mfma_f16x2_v4p", "mfma_f16x2_v4k"). */
#define SPML_ABI_VERSION 4
int spml_abi_version(void);

/* ------------------------------------------------------------------------
 * Deterministic mode (process-wide, off by default; 4: round 6).  The segment sums of A4 and the prototype
 * gradient of A9/A10 are sums of many fp32 terms that meet through atomics: run to run the bits differ (the reference's
 * own `scatter_add_` / `index_add` on a GPU does the same, segsort/common.py:34-39).  With the mode on, those sums are
 * formed in 64-bit fixed point (2^36 steps per unit, integer atomics: order-independent, hence bit-reproducible):
 *   - spml_segment_sum_normalize_f32 is refused (SPML_ERR_UNSUPPORTED); use spml_segment_sum_normalize_det_f32
 *     (same result to one fp32 rounding of the exact sum; domain |x| < 2^26 / P);
 *   - spml_segsort_nll_bwd_f32 accumulates d_protos / gscale in fixed point (its workspace grows by M * D * 8 bytes:
 *     query spml_segsort_nll_workspace_bytes AFTER switching the mode);
 *   - the generic k-means route does the same for its M-step (workspace: + n_img * K * D * 8 bytes);
 *   - spml_conv_hl8_pyramid_f32 does not split its taps over workgroups.
 * k-means fast paths, K1, the forward NLL, top-k, relabel and the convolutions are deterministic in either mode.
 * (Whole training steps: the Python side -- spml_amd/train.py -- additionally routes the framework convolutions that
 * remain through fixed-order GEMMs, spml_amd/nn/conv.py; tests/test_determinism_gpu.py.)
 * spml_set_deterministic returns the previous value.
 * ------------------------------------------------------------------------ */
int spml_set_deterministic(int on);
int spml_get_deterministic(void);

/* 0 for a product build.  Non-zero: the library was compiled with the profiling switches of csrc/conv.hip (low 16 bits,
 * SPML_CONV_EXP) or csrc/kmeans64.hip (high 16 bits, SPML_P64_EXP) -- variants that skip work and overwrite outputs;
 * spml_amd/_ffi.py refuses such a library unless the same environment variable is still set. */
int spml_build_experiment(void);

/* ------------------------------------------------------------------------
 * K1  normalise + NCHW->NHWC + location concat + normalise
 * replaces: segsort/common.py:306-310 (permute/contiguous/normalize),
 *           :313-317 (location features), :346-352 (cat + normalize),
 *           :355-365 (ignore-pixel removal, folded in through row_map),
 *           general/common.py:101-120 (normalize_embedding).
 *
 *   emb      [N,C,H,W]   input embedding map
 *   loc      [N,H,W,2]   location features, or NULL -> generated in-kernel as
 *                        (y/(H-1)-0.5, x/(W-1)-0.5)  (common.py:156-189 'float')
 *   row_map  [N*H*W]     destination row of every pixel, or -1 to drop it
 *                        (ignore_index pixels); NULL -> identity
 *   out_emb  [P',C]      x / max(|x|,1e-12)
 *   out_loc  [P',C+2]    normalize(cat(out_emb, loc))
 * backward: d_emb[N,C,H,W] from d_out_emb / d_out_loc (either may be NULL).
 * ------------------------------------------------------------------------ */
int spml_normalize_concat_loc_f32(const float* emb, int N, int C, int H, int W,
                                  const float* loc, const int64_t* row_map,
                                  float* out_emb, float* out_loc, void* stream);

int spml_normalize_concat_loc_bwd_f32(const float* emb, int N, int C, int H,
                                      int W, const float* loc,
                                      const int64_t* row_map,
                                      const float* d_out_emb,
                                      const float* d_out_loc, float* d_emb,
                                      void* stream);

/* K1 with L local-feature channels instead of the 2 location channels: local [N,H,W,L]
 * (1 <= L <= 8; the DensePose recipe appends (y, x) + 3 smoothed colours,
 * resnet_pspnet_densepose.py:37-38 -> L = 5); out_loc / d_out_loc are [P',C+L].
 * local == NULL is only allowed with L == 2 (location generated in-kernel). */
int spml_normalize_concat_local_f32(const float* emb, int N, int C, int H, int W,
                                    const float* local, int L,
                                    const int64_t* row_map, float* out_emb,
                                    float* out_loc, void* stream);

int spml_normalize_concat_local_bwd_f32(const float* emb, int N, int C, int H,
                                        int W, const float* local, int L,
                                        const int64_t* row_map,
                                        const float* d_out_emb,
                                        const float* d_out_loc, float* d_emb,
                                        void* stream);

/* The same two calls for a channels-last map (emb / d_emb stored [N, H, W, C], what the backbone hands
 * over when it runs NHWC): the pixel rows are already contiguous, K1 is a row-wise stream without a
 * transposition.  C in {16, 32, 64, 128, 256, 512} (spml_normalize_concat_local_nhwc_supported), else
 * SPML_ERR_UNSUPPORTED -- convert and use the NCHW entry points. */
int spml_normalize_concat_local_nhwc_supported(int C, int L);
int spml_normalize_concat_local_nhwc_f32(const float* emb, int N, int C, int H, int W,
                                         const float* local, int L,
                                         const int64_t* row_map, float* out_emb,
                                         float* out_loc, void* stream);
int spml_normalize_concat_local_nhwc_bwd_f32(const float* emb, int N, int C, int H,
                                             int W, const float* local, int L,
                                             const int64_t* row_map,
                                             const float* d_out_emb,
                                             const float* d_out_loc, float* d_emb,
                                             void* stream);

/* Plain row-wise L2 normalise of a [rows, D] matrix (general/common.py:101-120),
 * and its backward.  inv_norm_out (optional) receives 1/max(|x|,eps). */
int spml_normalize_rows_f32(const float* x, int64_t rows, int D, float* y,
                            void* stream);
int spml_normalize_rows_bwd_f32(const float* x, const float* dy, int64_t rows,
                                int D, float* dx, void* stream);

/* ------------------------------------------------------------------------
 * A7  dense re-indexing of integer keys (label algebra)
 * replaces: the `torch.unique(keys, return_inverse=True)` calls of
 *           segsort/common.py:192-218 (prepare_prototype_labels), :398-405 (segment_by_kmeans)
 *           and models/utils.py:94-111 (gather_clustering_and_update_prototypes)
 *   keys [P] int64 (any value except INT64_MIN) ->
 *   inv [P] int64: position of keys[i] among the sorted distinct keys;
 *   uniq [uniq_capacity] int64: the first uniq_capacity sorted distinct keys (pass P to get all);
 *   count [1] int64 (device): number of distinct keys.
 * Hash set + rank among the distinct keys: no sort of the P keys, deterministic, stream-ordered, no
 * host synchronisation (the caller reads `count` when it needs the number on the host).
 * ------------------------------------------------------------------------ */
size_t spml_relabel_unique_workspace_bytes(int64_t P);
int spml_relabel_unique_i64(const int64_t* keys, int64_t P, int64_t* inv, int64_t* uniq,
                            int64_t uniq_capacity, int64_t* count, void* workspace,
                            size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A3  grid initialisation of cluster labels
 * replaces: segsort/common.py:129-153 (initialize_cluster_labels)
 *   out [H,W] int64 = round(linspace(0,Ky-1,H))[:,None]
 *                     + (max_y+1) * round(linspace(0,Kx-1,W))[None,:]
 * ------------------------------------------------------------------------ */
int spml_kmeans_init_grid_i64(int H, int W, int Ky, int Kx, int64_t* out,
                              void* stream);

/* ------------------------------------------------------------------------
 * A4+A5+A6  spherical k-means over a ragged batch of images
 * replaces: segsort/common.py:67-97 (kmeans_with_initial_labels) and the
 *           per-image Python loop at :337-373, i.e. for every image
 *           `iterations` x { M-step :11-41 (scatter_add + normalize),
 *                            E-step :44-64 (mm + argmax) }.
 *
 *   x            [P,D]        unit-norm rows (embedding + location), P = sum of
 *                             all images' kept pixels, images back to back
 *   seg_offsets  [n_img+1]    DEVICE int64: image b owns rows
 *                             [seg_offsets[b], seg_offsets[b+1])
 *   max_seg_len               HOST upper bound on any image's row count (H*W)
 *   K                         clusters per image (labels are 0..K-1 per image)
 *   labels_init  [P]          int64 initial labels
 *   labels_out   [P]          int64 final labels (may alias labels_init)
 *   centroids_out [n_img,K,D] optional: the prototypes used by the last E-step
 *   flags                     SPML_KMEANS_* bits
 * One "pass" streams X once: E-step against the current prototypes fused with
 * the M-step accumulation for the next ones (iterations+1 passes in total, the
 * first one accumulate-only).  Results are run-to-run deterministic.
 * ------------------------------------------------------------------------ */
#define SPML_KMEANS_DEFAULT 0
#define SPML_KMEANS_FORCE_GENERIC 1 /* skip the MFMA fast paths (testing) */
#define SPML_KMEANS_FORCE_V2 4       /* use the 32x32-tile kernel even where v3 applies */
#define SPML_KMEANS_SEPARATE_PRECONVERT 16 /* convert X in its own kernel instead of inside
                                       the seed pass (testing / profiling) */
#define SPML_KMEANS_NO_PRECONVERT 8 /* keep X in fp32 and split it inside every pass (the
                                       default for < 3 passes) instead of converting it
                                       once to the MFMA operand layout up front */
#define SPML_KMEANS_NO_SCREEN 64     /* many-cluster path: skip the hi-half screening pass and score
                                       every pixel with the exact split-f16 kernel (testing / A-B) */
#define SPML_KMEANS_WS_PRECONVERTED 32 /* assign / fused pass: `ws` already holds X converted by
                                       spml_kmeans_preconvert_f32 (same x, sizes, ws) */
#define SPML_KMEANS_PASS_ONLY 256    /* spml_kmeans_fused_pass_f32 with SPML_KMEANS_WS_PRECONVERTED, measurement
                                       only: launch the pass kernel alone -- `ws` still holds the split centroids
                                       of a previous call with the same arguments, labels are written, the
                                       per-workgroup sums stay in `ws` and centroid_sums_out is not touched;
                                       bits 16..23 of flags: number of back-to-back launches (0 = 1) */

#define SPML_KMEANS_NO_PASS64 512    /* K <= 48 on pre-converted tiles: keep the E-step passes on the
                                       32-pixel-tile kernel (kmeans_pass16) instead of the pixel-split
                                       64-pixel-tile kernel (kmeans_pass64) (testing / A-B) */

#define SPML_KMEANS_TWO_KERNEL_FINALIZE 1024 /* slabs -> prototypes with kmeans_reduce_slabs + kmeans_normalize
                                       instead of the one-launch kmeans_finalize (testing / A-B).  NOTE on bits: a call
                                       is bit-reproducible run to run, but the two forms combine the per-workgroup
                                       partial sums in different (fixed) orders, and the default picks the one-launch
                                       form from 128 (padded cluster, image) rows on -- so the SAME image can get
                                       prototypes that differ in the last bit (and near-tie labels) depending on how
                                       many images share the call; pass this flag where the bits must not depend on
                                       the batch */

#define SPML_KMEANS_NO_V4K 2048      /* 64 < K <= 144 on wide rows (D = 128.. / 256.. + <= 8): skip the wave-split
                                       assign + accumulate passes on 64-pixel tiles ("mfma_f16x2_v4k"); the call
                                       takes the many-cluster kernels ("mfma_f16x2_bigk") (testing / A-B) */

size_t spml_kmeans_workspace_bytes(int64_t P, int D, int K, int n_img,
                                   int64_t max_seg_len);

/* Measurement aid (bench.py): one wave waits `spin_us` microseconds and writes out[0] = elapsed shader cycles
 * (s_memtime), out[1] = elapsed 100-MHz ticks (s_memrealtime): shader clock = 100 MHz * out[0] / out[1] while the
 * kernels of the other streams run. */
int spml_clock_probe(uint64_t* out, int spin_us, void* stream);

/* 3 x 3 / stride 2 / padding 1 max-pool of a channels-last map x [n][H][W][C] -> y [n][(H-1)/2+1][(W-1)/2+1][C]
 * (the stem's nn.MaxPool2d(kernel_size=3, stride=2, padding=1), spml/models/backbones/resnet.py:66-110).  Forward
 * only (conv1 / res2 are frozen: no gradient flows here).  C % 4 == 0, 16-byte aligned pointers. */
int spml_maxpool3x3s2_nhwc_f32(const float* x, int n, int H, int W, int C, float* y, void* stream);

int spml_kmeans_run_f32(const float* x, int64_t P, int D,
                        const int64_t* seg_offsets, int n_img,
                        int64_t max_seg_len, int K, const int64_t* labels_init,
                        int iterations, int64_t* labels_out,
                        float* centroids_out, int flags, void* ws,
                        size_t ws_bytes, void* stream);

/* E-step alone (find_nearest_prototypes, segsort/common.py:44-64) for a ragged
 * batch: labels_out[p] = argmax_k <x_p, centroids[img(p),k]>, ties -> lowest k.
 * centroids [n_img,K,D] need not be normalised.
 * Input domain of the MFMA paths: operands are split into two f16 halves (22 mantissa
 * bits, f16 exponent range), so every |x| and |centroid| element must be < 65504 and the
 * dot products are exact to ~2^-22 RELATIVE TO the operands' largest elements; unit-norm
 * rows (what the reference clusters) are the intended use.  The many-cluster path
 * ("mfma_f16x2_bigk") additionally needs |x| <= 1 in its M-step (fixed-point sums).  Pass
 * SPML_KMEANS_FORCE_GENERIC for arbitrary fp32 data (plain fp32 FMA kernels). */
int spml_kmeans_assign_f32(const float* x, int64_t P, int D,
                           const int64_t* seg_offsets, int n_img,
                           int64_t max_seg_len, int K, const float* centroids,
                           int64_t* labels_out, int flags, void* ws,
                           size_t ws_bytes, void* stream);

/* One fused pass (A5 + the scatter-sum half of A4; segsort/common.py:44-64 then :11-36):
 *   labels_out[p]        = argmax_k <x_p, centroids_in[img(p),k]>        int64 [P]
 *   centroid_sums_out    = sum of the rows of X by their NEW label       [n_img,K,D]
 * (un-normalised; the caller normalises, e.g. with spml_normalize_rows_f32, to get the
 * next prototypes).  X is streamed from HBM once.  This is the kernel the HBM roofline
 * of the k-means path is quoted on ("kmeans_pass16" for D = 32q + {0,2}, K <= 64).
 * With SPML_KMEANS_WS_PRECONVERTED the pass reads the split-f16 tiles that
 * spml_kmeans_preconvert_f32 left in `ws` (what spml_kmeans_run_f32 does from its second
 * pass on); without it X is split inside the pass. */
int spml_kmeans_fused_pass_f32(const float* x, int64_t P, int D,
                               const int64_t* seg_offsets, int n_img,
                               int64_t max_seg_len, int K,
                               const float* centroids_in, int64_t* labels_out,
                               float* centroid_sums_out, int flags, void* ws,
                               size_t ws_bytes, void* stream);

/* X fp32 [P,D] -> split-f16 fragment tiles inside `ws` (once per X; only for the shapes
 * whose pass kernels take pre-converted tiles, else SPML_ERR_UNSUPPORTED). */
int spml_kmeans_preconvert_f32(const float* x, int64_t P, int D,
                               const int64_t* seg_offsets, int n_img,
                               int64_t max_seg_len, int K, void* ws,
                               size_t ws_bytes, void* stream);

/* Name of the code path a k-means call with these (host-visible) arguments takes; a pure
 * function (no state), for tests and the bench report.  given_centroids != 0: the assign /
 * fused-pass entry points.
 *   "mfma_f16x2_v4p"  D = 32q + {0,2}, q in {1,2,4,8}, K <= 48, >= 3 passes (pre-converted X): E-step passes
 *                     on kmeans_pass64 (64-pixel tiles, pixel-split E-step), seed pass on kmeans_pass16
 *   "mfma_f16x2_v3p"  the same for 48 < K <= 64 (or SPML_KMEANS_NO_PASS64): every pass on kmeans_pass16
 *   "mfma_f16x2_v3"   same shapes, < 3 passes (tile split to f16 in LDS inside the pass)
 *   "mfma_f16x2_v3k"  64 < K <= 256, q in {1,2,4} within the register budget, >= 3 passes
 *   "mfma_f16x2_v4k"  64 < K <= 144, q in {4,8}, tail <= 8 outside v3k (e.g. K = 144, D = 258), >= 3 passes: an assign
 *                     kernel with the prototype tiles split over eight waves + an accumulate kernel on the same
 *                     64-pixel tiles (kmeans64k.hip); SPML_KMEANS_NO_V4K: "mfma_f16x2_bigk"
 *   "mfma_f16x2"      other even D <= 320 with K <= 64 (32x32x16 tiles, k-split)
 *   "mfma_f16x2_bigk" K > 64 outside the shapes above with D <= 528 (e.g. K = 1024, D = 514;
 *                     kmeans_big.hip): pixel-stationary MFMA E-step with a running arg-max,
 *                     counting-sort + fixed-point gather M-step
 *   "generic"         everything else (fp32 FMA assign + scatter-sum) */
const char* spml_kmeans_path_name(int64_t P, int D, int K, int n_img,
                                  int64_t max_seg_len, int iterations,
                                  int given_centroids, int flags);

/* Profiling variant of spml_kmeans_run_f32 (tile-kernel paths only): every workgroup of
 * every pass kernel stamps its start and end time (s_memrealtime, 100 MHz) into
 * pass_clocks[pass][workgroup][2] -- device memory, nothing is synchronised and no host
 * timers or events are involved; the caller derives each launch's duration as
 * max(end) - min(start).  Pass p of iterations+1: 0 = seed (M-step on labels_init),
 * 1..iterations-1 = fused E+M passes, iterations = final E-only pass.
 * spml_kmeans_profile_layout gives the two dimensions. */
int spml_kmeans_profile_layout(int64_t P, int D, int K, int n_img,
                               int64_t max_seg_len, int iterations,
                               int* n_passes, int* workgroups_per_pass);

int spml_kmeans_run_profiled_f32(const float* x, int64_t P, int D,
                                 const int64_t* seg_offsets, int n_img,
                                 int64_t max_seg_len, int K,
                                 const int64_t* labels_init, int iterations,
                                 int64_t* labels_out, int flags, void* ws,
                                 size_t ws_bytes, uint64_t* pass_clocks,
                                 size_t pass_clocks_len, void* stream);

/* ------------------------------------------------------------------------
 * A4  segment prototypes: scatter-sum rows by id, then L2 normalise
 * replaces: segsort/common.py:11-41 (calculate_prototypes_from_labels) as
 *           used at models/utils.py:113-116 and segsort.py:236-237.
 *   x [P,D], ids [P] int64 in [0,M)  ->  protos [M,D]; sums [M,D] is scratch
 *   that also feeds the backward (it holds the un-normalised sums).
 * backward: dx[p] = J(ids[p]) where J = d_sums = (dP - P<P,dP>)/|s|
 *           (or dP/eps where |s| < eps).  `accumulate` != 0 adds into dx.
 * ------------------------------------------------------------------------ */
int spml_segment_sum_normalize_f32(const float* x, const int64_t* ids,
                                   int64_t P, int D, int64_t M, float* sums,
                                   float* protos, void* stream);

/* deterministic form (see spml_set_deterministic): ws of spml_segment_sum_det_workspace_bytes(M, D) bytes */
size_t spml_segment_sum_det_workspace_bytes(int64_t M, int D);
int spml_segment_sum_normalize_det_f32(const float* x, const int64_t* ids,
                                       int64_t P, int D, int64_t M,
                                       float* sums, float* protos, void* ws,
                                       size_t ws_bytes, void* stream);

int spml_segment_sum_normalize_bwd_f32(const float* d_protos,
                                       const float* sums, const int64_t* ids,
                                       int64_t P, int D, int64_t M,
                                       float* d_sums_scratch, float* dx,
                                       int accumulate, void* stream);

/* ------------------------------------------------------------------------
 * A9/A10  pixel-to-segment NCA negative log-likelihood ('segsort+' mode)
 * replaces: segsort/loss.py:15-82 (_calculate_log_likelihood) and :85-130
 *           (_one_hot_calculate_log_likelihood): mm -> *kappa -> exp ->
 *           gather own column -> label masks -> two masked row sums ->
 *           where / divide / log.
 *
 *   emb [P,D], protos [M,D] fp32; own [P] int64 = index of the pixel's own
 *   segment among the M prototypes.
 *   mode SPML_NLL_LABEL : positives have  px_code[p] == pr_code[m]
 *   mode SPML_NLL_TAGSET: positives have (px_code[p] &  pr_code[m]) != 0
 *        (multi-hot tag sets packed into 64-bit masks; loss.py:95-113 computes
 *        the same predicate as `tags_px @ tags_pr.T > 0`)
 *   outputs: nll [P]; stats [P,4] = (num, den, own_sim, fallback flag) kept
 *   for the backward.
 * backward: given d_nll [P] (upstream gradient per pixel) produces
 *   d_emb [P,D] and d_protos [M,D] (d_protos is ADDED into; zero it first).
 *   Only prototypes [0, m_grad) receive a gradient (m_grad < 0: all M) -- rows of
 *   a detached memory bank appended after the live prototypes are skipped.
 * ------------------------------------------------------------------------ */
#define SPML_NLL_LABEL 0
#define SPML_NLL_TAGSET 1
/* OR-able: group_mode != 'segsort+' (loss.py:71-72): numerator = own-segment
 * similarity only, no positive set */
#define SPML_NLL_PLAIN 2
/* OR-able promise: every px_code / pr_code value fits in 32 bits (labels in [0, 2^31); tag sets
 * over <= 32 classes).  The predicate then runs on 32-bit words (the kernels are bound by
 * that per-pair epilogue); results are identical. */
#define SPML_NLL_CODE32 4

size_t spml_segsort_nll_workspace_bytes(int64_t P, int64_t M, int D);

int spml_segsort_nll_fwd_f32(const float* emb, const int64_t* own,
                             const int64_t* px_code, int64_t P,
                             const float* protos, const int64_t* pr_code,
                             int64_t M, int D, float kappa, int mode,
                             float* nll, float* stats, void* ws,
                             size_t ws_bytes, void* stream);

int spml_segsort_nll_bwd_f32(const float* emb, const int64_t* own,
                             const int64_t* px_code, int64_t P,
                             const float* protos, const int64_t* pr_code,
                             int64_t M, int D, float kappa, int mode,
                             const float* stats, const float* d_nll,
                             float* d_emb, float* d_protos, int64_t m_grad,
                             void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * A11/B3  top-k retrieval by cosine affinity
 * replaces: segsort/eval.py:32-35 (mm + full argsort + [:, :k]) and
 *           models/utils.py:198-214 (mm + where(mask) + topk).
 *   q [Q,D], protos [M,D]  ->  idx [Q,k] int64 (descending affinity, ties ->
 *   lowest index), val [Q,k] fp32.  k <= 32.
 *   Optional mask predicate (B3): candidate m is allowed for query i iff
 *   q_group[i] == pr_group[m] && pr_valid[m] != 0; disallowed candidates rank
 *   after every allowed one with value `masked_value`.  Pass NULLs for none.
 * ------------------------------------------------------------------------ */
size_t spml_topk_workspace_bytes(int64_t Q, int64_t M, int D, int k);

int spml_topk_affinity_f32(const float* q, int64_t Q, const float* protos,
                           int64_t M, int D, int k, const int64_t* q_group,
                           const int64_t* pr_group, const uint8_t* pr_valid,
                           float masked_value, int64_t* idx, float* val,
                           void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * N2  overlap accumulation of sliding-window crop embeddings (full-resolution
 *     inference, SURVEY.md 8f "next" row N2)
 * replaces: pyscripts/inference/prototype.py:163-178 (the same code is in
 *           inference.py:175-200): per crop, normalize_embedding over the
 *           channels, `embeddings[:, :, sh:eh, sw:ew] += crop`, `counts += 1`.
 *   patch  [C,h,w]   crop embedding of ONE image (NCHW, N = 1)
 *   acc    [C,H,W]   full-resolution sum, updated in place
 *   counts [H,W]     number of crops covering each pixel, updated in place
 *   (sh, sw)         top-left corner of the crop inside the full map
 * The caller divides acc by counts once all crops are in (prototype.py:180-181).
 * ------------------------------------------------------------------------ */
int spml_window_accumulate_f32(const float* patch, int C, int h, int w,
                               float* acc, float* counts, int H, int W, int sh,
                               int sw, void* stream);

/* ------------------------------------------------------------------------
 * N3  pixel x pixel affinity -> random-walk transition matrix (pseudo labels,
 *     SURVEY.md 8f "next" row N3)
 * replaces: pyscripts/inference/pseudo_camrw_crf.py:146-158 (the same code is in
 *           pseudo_softmaxrw_crf.py:137-163): per view
 *           `matmul(E^T, E).mul_(5).add_(-5).exp_()`, mean over the views,
 *           `** 20`, `/ sum(dim=0)`.
 *   emb    [B,C,n]  B augmented views of one image, columns already unit-norm
 *                   (n = (H/8)*(W/8) pixels, C channels)
 *   trans  [n,n]    (mean_b exp(scale*<e_i,e_j> - scale))^power / column sums
 * The walk (trans <- trans @ trans, x6) and `cam @ trans` stay library GEMMs.
 * ------------------------------------------------------------------------ */
size_t spml_affinity_workspace_bytes(int B, int C, int64_t n);

int spml_affinity_transition_f32(const float* emb, int B, int C, int64_t n,
                                 float scale, int power, float* trans, void* ws,
                                 size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------
 * G1 (backbone)  training-mode batch normalisation fused with the ReLU and the residual add
 * replaces: the framework op chains of the reference's bottleneck unit,
 *           spml/models/backbones/resnet.py:42-63  (`relu(bn(conv(x)))` twice, then
 *           `relu(bn(conv(x)) + identity)`), and of its stem (:66-110), for channels-last
 *           fp32 activations.
 *   x, residual, y, dy, dx, d_residual   [R, C] rows = N*H*W pixels (NHWC), C % 4 == 0
 *   mean, m2, invstd, gamma, beta, sums  [C]
 * forward:  spml_bn_stats_f32 -> (the caller turns m2 into invstd = rsqrt(m2 / count + eps),
 *           combining ranks first for SyncBatchNorm, and updates the running statistics) ->
 *           spml_bn_act_apply_f32:  y = act((x - mean) * invstd * gamma + beta [+ residual]).
 * backward: spml_bn_act_bwd_reduce_f32 -> sum_dz = sum(dz), sum_dz_xhat = sum(dz * xhat) with
 *           dz = dy * (y > 0) (y == NULL: no ReLU), the gradients of beta / gamma ->
 *           (all-reduce for SyncBatchNorm) -> spml_bn_act_bwd_apply_f32:
 *           dx = gamma * invstd * (dz - sum_dz / count - xhat * sum_dz_xhat / count),
 *           d_residual = dz.  dx or d_residual may be NULL.  `count` is the number of rows the
 *           statistics were pooled over: a host value, or -- SyncBatchNorm, where the ranks' row
 *           counts may differ (lib/nn/sync_batchnorm/batchnorm.py:124-145 sums the gathered
 *           sizes) -- the device float `count_dev` that spml_bn_finalize_ranks_f32 wrote
 *           (non-NULL count_dev wins; no host read of the gathered counts).
 * ------------------------------------------------------------------------ */
size_t spml_bn_workspace_bytes(int64_t R, int C);

/* Single-rank batch norm, everything in one call: statistics -> invstd + running statistics
 * (momentum; unbiased variance, as torch.nn.BatchNorm2d) -> apply.  mean / invstd [C] are kept
 * for the backward call, which returns d_gamma = sum(dz * xhat), d_beta = sum(dz), dx and
 * (optionally) d_residual. */
int spml_bn_act_fwd_f32(const float* x, const float* residual, int64_t R, int C,
                        const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum,
                        float eps, int relu, float* y, float* mean, float* invstd,
                        void* ws, size_t ws_bytes, void* stream);

int spml_bn_act_bwd_f32(const float* dy, const float* y, const float* x, int64_t R,
                        int C, const float* mean, const float* invstd,
                        const float* gamma, float* d_gamma, float* d_beta,
                        float* dx, float* d_residual, void* ws, size_t ws_bytes,
                        void* stream);

int spml_bn_stats_f32(const float* x, int64_t R, int C, float* mean, float* m2,
                      void* ws, size_t ws_bytes, void* stream);

int spml_bn_act_apply_f32(const float* x, const float* residual, int64_t R, int C,
                          const float* mean, const float* invstd,
                          const float* gamma, const float* beta, int relu,
                          float* y, void* stream);

int spml_bn_act_bwd_reduce_f32(const float* dy, const float* y, const float* x,
                               int64_t R, int C, const float* mean,
                               const float* invstd, float* sum_dz,
                               float* sum_dz_xhat, void* ws, size_t ws_bytes,
                               void* stream);

int spml_bn_act_bwd_apply_f32(const float* dy, const float* y, const float* x,
                              int64_t R, int C, const float* mean,
                              const float* invstd, const float* gamma,
                              const float* sum_dz, const float* sum_dz_xhat,
                              double count, const float* count_dev, float* dx,
                              float* d_residual, void* stream);

/* ---- stride-1 convolutions of the bottleneck stack on the f16 matrix cores, fp32-class ----
 * Replaces the framework convolutions of spml/models/backbones/resnet.py:20-33,42-63 (conv1 /
 * conv2 / conv3 / downsample of a Bottleneck with stride 1: every unit of res4 and res5).
 * "hl8" = split-f16 copy of an fp32 tensor [rows][C] (C % 8 == 0), rows*C*4 bytes:
 *   v * S = h + l, 16-byte units ((row*C/8 + c/8)*2 + part) of 8 channels, part 0 = h, 1 = l;
 *   S = 2^(14-e) for the smallest e with *bound < 2^e (S = 1 when bound is NULL); *bound is a
 *   device float >= max|v|.                                                            */

/* x fp32 [rows][C] -> hl8.  compute_bound != 0: *bound = max|x| is computed first (one more
 * pass over x); otherwise *bound (may be NULL) is taken as given. */
int spml_hl8_from_f32(const float* x, int64_t rows, int C, float* bound,
                      int compute_bound, void* out, void* stream);

/* weights fp32 [Cout][taps][Cin] (the channels-last storage of a conv weight) -> hl8
 * [Cin][taps][Cout] with mirrored taps: the B operand of the data-gradient convolution. */
int spml_hl8_weight_transposed_f32(const float* w, int Cout, int taps, int Cin,
                                   const float* bound, void* out, void* stream);

/* 1 when spml_conv_hl8_f32 handles (K input channels, N output channels, taps in {1, 9}):
 * K % 16 == 0 and N % 64 == 0.  N % 256 == 0 runs the 256-column tiles the kernel is tuned for;
 * N % 128 == 0 and N % 64 == 0 run 128- / 64-column tiles (2 x 2 and 1 x 4 waves), which re-read
 * the activations more often per output (measured 1.5x the library on the res3 shapes, slower than
 * it on the 2048 -> 64 ASPP branches: profiles/r02_conv_kernels.md). */
int spml_conv_hl8_supported(int K, int N, int taps);

/* out[r][n] = sum_{tap,k} a[r + shift(tap)][k] * b[n][tap][k]  (+ addend[r][n]),
 * r = (img*H + oh)*W + ow, shift(tap) = ((tap/3-1)*W + (tap%3-1)) * dilation with zero padding
 * (taps == 9) or 0 (taps == 1).  a: hl8 [n_img*H*W][K], b: hl8 [N][taps*K], out fp32 NHWC.
 * Forward: a = activations, b = weights.  Data gradient: a = dy, b = transposed weights,
 * K = Cout, N = Cin, addend = gradient of the residual branch or NULL; with addend_mask (the ReLU
 * mask bytes of spml_bn_*: [R][N/4], bit n & 3) the addend is the unit's OUTPUT gradient and
 * addend * mask = the residual branch's gradient is formed here instead of being written by the
 * batch-norm backward. */
int spml_conv_hl8_f32(const void* a, const float* a_bound, const void* b,
                      const float* b_bound, const float* addend,
                      const unsigned char* addend_mask, float* out,
                      int n_img, int H, int W, int K, int N, int taps, int dilation,
                      void* stream);

/* The same convolution when a training-mode batch norm follows (resnet.py:42-63: every convolution of a
 * Bottleneck): the epilogue also leaves, per row tile of the launch and per output channel, the mean, the
 * sum of squared deviations, the max and the min of `out` in chunk_stats [4][chunks][N] -- what
 * spml_bn_fwd_hl8_chunks_f32 pools instead of reading `out` once more -- and resets *zero_me (that call's
 * y_bound; may be NULL).  spml_conv_hl8_stats_layout gives chunks / chunk_rows for a shape, or
 * SPML_ERR_UNSUPPORTED where the tiling has no per-tile column owner (N % 256 != 0). */
int spml_conv_hl8_stats_layout(int n_img, int H, int W, int K, int N, int taps,
                               int* chunks, int* chunk_rows);
int spml_conv_hl8_stats_f32(const void* a, const float* a_bound, const void* b,
                            const float* b_bound, float* out, float* chunk_stats,
                            float* zero_me, int n_img, int H, int W, int K, int N,
                            int taps, int dilation, void* stream);

/* Weight gradient of the same convolutions:
 *   dw[n][tap][k] = sum_r dy[r][n] * x[r + shift(tap)][k]      (dw fp32 [N][taps][K] = the
 * channels-last storage of the weight gradient).  dy: hl8 [R][N], x: hl8 [R][K] (the forward
 * input).  K % 128 == 0 and N % 128 == 0 (256-wide output tiles where the channel count allows it,
 * 128-wide ones otherwise).  The pixel range is split over workgroups; the
 * partial tiles live in the caller's workspace and are summed in a fixed order. */
int spml_conv_wgrad_hl8_supported(int K, int N, int taps);
size_t spml_conv_wgrad_workspace_bytes(int n_img, int H, int W, int K, int N, int taps);
int spml_conv_wgrad_hl8_f32(const void* dy, const float* dy_bound, const void* x,
                            const float* x_bound, float* dw, int n_img, int H, int W,
                            int K, int N, int taps, int dilation, void* ws,
                            size_t ws_bytes, void* stream);

/* Weight gradients of the pyramid head's narrow branches (spml/models/heads/spp.py:8-43: up to four
 * dilated 3x3 convolutions of ONE input whose outputs are summed -- they share dy): N == 64,
 * K % 256 == 0, 1..4 branches with dilations[b] in 1..255 (host array).
 *   dw[b][n][tap][k] = sum_r dy[r][n] * x[r + shift_b(tap)][k]
 * dw fp32 [branches][64][9][K] = the channels-last storage of each branch's [64, K, 3, 3] weight
 * gradient, one branch after the other.  x is streamed once per FOUR taps (a 256-column tile is
 * four taps x 64 channels), not once per tap.  Workspace as above. */
int spml_conv_wgrad_pyramid_hl8_supported(int K, int N, int branches);
/* Forward of the same narrow branches as ONE 1x1 convolution + a gather: with
 *   z[r][tap * N + n] = x[r] . w_tap[n]        (spml_conv_hl8_f32 with 9 * branches * N columns,
 *                                                tap = 9 * branch + 3 * kh + kw)
 * the head's output is  out[p][n] = bias[n] + sum_tap z[p + shift(tap)][tap * N + n]  (taps whose
 * shifted pixel leaves the image contribute 0), which this call gathers (taps added in order:
 * deterministic).  z fp32 [R][9 * branches * N], out fp32 [R][N]; N a power of two in 16..1024,
 * bias [N] or NULL, dilations[b] in 1..255 (host array). */
int spml_conv_tap_gather_f32(const float* z, const float* bias, float* out, int n_img, int H,
                             int W, int N, int branches, const int* dilations, void* stream);
size_t spml_conv_wgrad_pyramid_workspace_bytes(int n_img, int H, int W, int K, int N,
                                               int branches);
int spml_conv_wgrad_pyramid_hl8_f32(const void* dy, const float* dy_bound, const void* x,
                                    const float* x_bound, float* dw, int n_img, int H, int W,
                                    int K, int N, int branches, const int* dilations,
                                    void* ws, size_t ws_bytes, void* stream);

/* ---- batch norm producing the split-f16 ("hl8") copies the matrix-core convolutions read ----
 * Same math as the spml_bn_* calls above (spml/models/backbones/resnet.py:42-63); y / dx can be
 * written as fp32, as hl8, or both.  The tensor bounds that fix the hl8 scales come from
 * per-channel extremes gathered in the statistics / reduction pass (no extra pass over the
 * data).  C % 8 == 0.  Cross-rank statistics are combined by the caller between the halves. */
int spml_bn_stats_ext_f32(const float* x, int64_t R, int C, float* mean, float* m2,
                          float* cmax, float* cmin, void* ws, size_t ws_bytes,
                          void* stream);

/* spml_bn_stats_ext_f32 pooled from the chunk statistics of the producing convolution
 * (spml_conv_hl8_stats_f32: chunk_stats [4][chunks][C]) instead of a pass over x -- the local
 * half of SyncBatchNorm for the matrix-core units. */
int spml_bn_stats_ext_chunks_f32(const float* chunk_stats, int chunks, int chunk_rows,
                                 int64_t R, int C, float* mean, float* m2, float* cmax,
                                 float* cmin, void* stream);

/* invstd = rsqrt(m2/count + eps); running statistics updated in place (may be NULL). */
int spml_bn_finalize_f32(const float* mean, const float* m2, int C, double count,
                         float eps, float momentum, float* running_mean,
                         float* running_var, float* invstd, void* stream);

/* y (fp32, may be NULL) and/or y_hl8 = act((x-mean)*invstd*gamma + beta [+ residual]);
 * *y_bound (required with y_hl8, optional otherwise) receives max_c |bn(x)_c| (+ *residual_bound);
 * relu_mask (may be NULL): one byte per (row, channel quad), bit e = (y[row][4q+e] > 0) -- what
 * the backward calls read instead of y. */
int spml_bn_act_apply_hl8_f32(const float* x, const float* residual,
                              const float* residual_bound, int64_t R, int C,
                              const float* mean, const float* invstd,
                              const float* gamma, const float* beta,
                              const float* cmax, const float* cmin, int relu,
                              float* y, void* y_hl8, float* y_bound,
                              unsigned char* relu_mask, void* stream);

/* ReLU mask from y (fp32) or from relu_mask (or none); also max |dz| per channel. */
int spml_bn_act_bwd_reduce_ext_f32(const float* dy, const float* y,
                                   const unsigned char* relu_mask,
                                   const float* x, int64_t R, int C, const float* mean,
                                   const float* invstd, float* sum_dz,
                                   float* sum_dz_xhat, float* max_dz, void* ws,
                                   size_t ws_bytes, void* stream);

int spml_bn_act_bwd_apply_hl8_f32(const float* dy, const float* y,
                                  const unsigned char* relu_mask,
                                  const float* x, int64_t R, int C, const float* mean,
                                  const float* invstd, const float* gamma,
                                  const float* sum_dz, const float* sum_dz_xhat,
                                  const float* max_dz, const float* cmax,
                                  const float* cmin, double count,
                                  const float* count_dev, float* dx, void* dx_hl8,
                                  float* dx_bound, float* d_residual, void* stream);

/* Single-rank forms (no cross-rank statistics): the whole forward / backward of one batch norm
 * in one call each, three launches.  mean / invstd / cmax / cmin are outputs of the forward and
 * inputs of the backward; d_gamma / d_beta are the parameter gradients. */
int spml_bn_fwd_hl8_f32(const float* x, const float* residual, const float* residual_bound,
                        int64_t R, int C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, float momentum, float eps,
                        int relu, float* y, void* y_hl8, float* y_bound,
                        unsigned char* relu_mask, float* mean, float* invstd, float* cmax,
                        float* cmin, void* ws, size_t ws_bytes, void* stream);

/* spml_bn_fwd_hl8_f32 from chunk statistics left by the producer of x (spml_conv_hl8_stats_f32):
 * chunk_stats [4][chunks][C] = mean, M2, max, min per chunk of chunk_rows rows (the last chunk
 * shorter); *y_bound must have been reset by the producer.  Two launches, x is read once. */
int spml_bn_fwd_hl8_chunks_f32(const float* x, const float* chunk_stats, int chunks,
                               int chunk_rows, const float* residual,
                               const float* residual_bound, int64_t R, int C,
                               const float* gamma, const float* beta, float* running_mean,
                               float* running_var, float momentum, float eps, int relu,
                               float* y, void* y_hl8, float* y_bound,
                               unsigned char* relu_mask, float* mean, float* invstd,
                               float* cmax, float* cmin, void* stream);

int spml_bn_bwd_hl8_f32(const float* dy, const float* y, const unsigned char* relu_mask,
                        const float* x, int64_t R, int C, const float* mean,
                        const float* invstd, const float* gamma, const float* cmax,
                        const float* cmin, float* d_gamma, float* d_beta, float* dx,
                        void* dx_hl8, float* dx_bound, float* d_residual, void* ws,
                        size_t ws_bytes, void* stream);

/* Sum of up to four 3x3 convolutions with different dilations over the same input in one launch
 * (the data gradient of the DeepLab-v2 ASPP head, spml/models/heads/spp.py:8-43: four dilated
 * branches whose outputs are summed, so all four see the same output gradient):
 *   out[r][n] = sum_g sum_{tap,k} a[r + shift_g(tap)][k] * b[n][(9 g + tap) * K + k]  (+ bias[n]) (+ addend)
 * -- with the forward weights concatenated along the taps it is also the head's forward pass.
 * b: hl8 [N][9 * groups * K]; spml_hl8_weight_transposed_into_f32 writes one branch's mirrored,
 * transposed weight into its tap range of that operand; spml_absmax_bound_f32 accumulates the
 * shared bound (max over calls unless zero_first). */
int spml_conv_hl8_pyramid_f32(const void* a, const float* a_bound, const void* b,
                              const float* b_bound, const float* bias, const float* addend,
                              float* out,
                              int n_img, int H, int W, int K, int N, int groups,
                              const int* dilations, void* stream);
int spml_hl8_weight_transposed_into_f32(const float* w, int Cout, int taps, int Cin,
                                        const float* bound, void* out, int taps_total,
                                        int tap_offset, void* stream);
int spml_absmax_bound_f32(const float* x, int64_t n, float* bound, int zero_first,
                          void* stream);

/* All conv weights of one bottleneck unit (n <= 4) in two launches: bounds[i] = max|w_i|, then
 * fwd[i] = hl8 [Cout][taps*Cin] and transposed[i] = hl8 [Cin][taps*Cout] (mirrored taps) of every
 * weight.  The pointer / size arrays are host arrays; bounds is a device array of 4 floats. */
int spml_hl8_weight_set_f32(const float* const* w, const int* cout, const int* cin,
                            const int* taps, int n, float* bounds, void* const* fwd,
                            void* const* transposed, void* stream);

/* SyncBatchNorm: stats [world][3][C] = the ranks' (count, mean, M2) as gathered by the caller ->
 * pooled mean, invstd = rsqrt(M2/count + eps), running statistics updated in place (may be NULL);
 * total_count [1] (may be NULL) receives the pooled row count = sum of the ranks' counts, the
 * `count_dev` of the backward apply calls. */
int spml_bn_finalize_ranks_f32(const float* stats, int world, int C, float eps,
                               float momentum, float* running_mean, float* running_var,
                               float* mean, float* invstd, float* total_count, void* stream);

/* Inference form of the same convolution (batch norm folded into weights and bias by the caller,
 * spml/models/backbones/resnet.py:42-63 in eval mode):
 *   out = act(conv(a, b) + bias[n] [+ addend]),   *out_bound = max |out|   (both optional)
 * so that the next convolution's hl8 input is one conversion pass away (spml_hl8_from_f32 with the
 * bound given). */
int spml_conv_hl8_affine_f32(const void* a, const float* a_bound, const void* b,
                             const float* b_bound, const float* bias, const float* addend,
                             int relu, float* out, float* out_bound, int n_img, int H,
                             int W, int K, int N, int taps, int dilation, void* stream);

/* ---- softmax head: cross-entropy of bilinearly up-sampled logits -------------------------
 * Replaces `F.interpolate(logits, size=labels.shape[-2:], mode='bilinear')` followed by
 * `CrossEntropyLoss(ignore_index)` in spml/models/predictions/segsort_softmax.py:112-131 (the
 * classifier head trained beside the contrastive terms) without materialising the [N, C, H, W]
 * logits.  logits: fp32 [N][h][w][C] (channels-last storage of the [N, C, h, w] map), C <= 64;
 * labels: int64 [N][H][W], every value in [0, C) or == ignore_index.
 *   fwd: lse [N][H][W] = logsumexp of each pixel's interpolated logits (kept for the backward pass),
 *        result[0] = sum of the pixel losses, result[1] = number of counted pixels,
 *        result[2] = their mean (NaN without counted pixels, like the framework loss);
 *   bwd: d_logits [N][h][w][C] = scale[0] * d(sum of the pixel losses) / d logits, scale a device
 *        scalar (d_loss / result[1] for the mean).  Gather formulation: deterministic, no atomics.
 * Interpolation arithmetic as ATen's upsample_bilinear2d (align_corners = False). */
int spml_upsample_ce_supported(int C);
size_t spml_upsample_ce_workspace_bytes(int N, int H, int W);
int spml_upsample_ce_fwd_f32(const float* logits, const int64_t* labels, int N, int C, int h,
                             int w, int H, int W, int64_t ignore_index, float* lse,
                             float* result, void* ws, size_t ws_bytes, void* stream);
int spml_upsample_ce_bwd_f32(const float* logits, const int64_t* labels, const float* lse,
                             int N, int C, int h, int w, int H, int W, int64_t ignore_index,
                             const float* scale, float* d_logits, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPML_HIP_H_ */
