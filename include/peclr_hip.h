/*
 * peclr_hip.h -- C ABI of libpeclr_hip.so: the MI355X (gfx950) kernels of the PeCLR
 * pretraining hot path.
 *
 * The reference (dahiyaaneesh/peclr) is pure Python and has NO FFI of its own: every op on
 * its hot path is a chain of stock PyTorch ops.  This header is therefore the boundary a
 * native replacement exports, one entry point per op chain of SURVEY.md section 2.2
 * (K1..K8); each declaration cites the reference lines it replaces (paths relative to the
 * reference root).  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates
 *     or frees memory; scratch buffers are caller-provided (sized with the *_jsplit /
 *     *_pick_split_k policy queries).  Buffers are row-major, contiguous, 16-byte aligned.
 *   - one entry point = one kernel launch (so a HIP-event pair around a call times exactly
 *     the kernel that rocprofv3 reports).
 *   - every call is asynchronous on the given hipStream_t (`stream`), re-entrant, keeps no
 *     global mutable state, and may be called concurrently on different streams.
 *   - return value: 0 = ok; negative = argument error (PECLR_ERR_*); positive = hipError_t
 *     of the failed launch.  No exceptions cross the ABI.
 *   - "slabs": a split-K producer writes S partial results [S][rows][cols]; the consuming
 *     kernel sums them on load (fused reduction).  S == 1 means a plain dense tensor.
 *   - row layout of two-view tensors: rows [0,N) are view-1 samples, rows [N,2N) view-2
 *     samples (hybrid2_model.py:30-32).  Multi-GPU gathered tensors repeat that block per
 *     rank; `n_half` (= N per rank) defines the positive-pair map: partner(i) = i + n_half
 *     if (i / n_half) is even else i - n_half.
 */
#ifndef PECLR_HIP_H
#define PECLR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* peclr_stream_t; /* hipStream_t */

#define PECLR_OK 0
#define PECLR_ERR_NULL (-1)        /* required pointer is null */
#define PECLR_ERR_SHAPE (-2)       /* unsupported / inconsistent shape */
#define PECLR_ERR_ALIGN (-3)       /* pointer or leading dimension not 16-byte aligned */
#define PECLR_ERR_WORKSPACE (-4)   /* workspace too small */
#define PECLR_ERR_UNSUPPORTED (-5) /* unsupported flag / dtype */

/* GEMM operand layouts: C[M,N] = A . B  with                                              */
#define PECLR_GEMM_NT 0 /* A[M,K] row-major, B[N,K] row-major  (y = x W^T : nn.Linear fwd) */
#define PECLR_GEMM_NN 1 /* A[M,K] row-major, B[K,N] row-major  (dx = dy W)                 */
#define PECLR_GEMM_TN 2 /* A[K,M] row-major, B[K,N] row-major  (dW = dy^T x)               */

/* align flags */
#define PECLR_ALIGN_CROP 1        /* "crop" in config.augmentation   (hybrid2_model.py:58)  */
#define PECLR_ALIGN_ROTATE 2      /* "rotate" in config.augmentation (hybrid2_model.py:76)  */
#define PECLR_ALIGN_SINGLE_NORM 4 /* SimCLR.contrastive_step: one F.normalize, no alignment
                                     (simclr_model.py:44-47)                                */

int peclr_version(void);
const char* peclr_error_string(int code);
/* Identity of the hipGraph capture `stream` is in (*id_out = 0: not capturing).  Plumbing for the host side, which packs
 * the six-product GEMMs' weight planes once per optimiser step AND once per capture (no reference counterpart: the
 * reference launches eagerly, peclr_training.py:96). */
int peclr_stream_capture_id(void* stream, unsigned long long* id_out);

/* ---- K1 / K2-GEMM and their backward GEMMs -------------------------------------------
 * Replaces nn.Linear forward/backward of the projection head (simclr_model.py:22-26,29-33).
 * fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32 (bit-exact fmaf chain).
 * split_k == 1: C = A.B (+ bias[N] if non-null).  split_k > 1: slabs[s] (each [M,N], ld = N)
 * receive the K-partials, C and bias are ignored (the consumer adds the bias).
 * Feature dimensions (the contiguous dimension of each operand, and N) must be multiples
 * of 4; M (batch rows) is arbitrary.                                                      */
int peclr_gemm_f32(int layout, int M, int N, int K, const float* A, int lda, const float* B,
                   int ldb, float* C, int ldc, const float* bias, int split_k, float* slabs,
                   peclr_stream_t stream);
/* The library's own split-K policy for a given problem (>= 1). */
/* C = op(A) op(B) + addend (same layouts as peclr_gemm_f32, no split-K): the 1x1-convolution input
 * gradient of a bottleneck's first conv with the residual branch's gradient added in the epilogue,
 * dX[R,Cin] = dY[R,Cmid] W[Cmid,Cin] + dRes[R,Cin] -- what autograd otherwise does as MIOpen dgrad +
 * a separate elementwise add over the block input (torchvision Bottleneck, resnet_model.py:15). */
int peclr_gemm_add_f32(int layout, int M, int N, int K, const float* A, int lda, const float* B,
                       int ldb, float* C, int ldc, const float* addend, int ldd,
                       peclr_stream_t stream);
/* The same for bf16 (autocast) backbones: C (bf16) = A[M,K] (bf16) . B[N,K]^T (bf16) + addend (bf16,
 * nullable), fp32 accumulation on v_mfma_f32_32x32x16_bf16; both operands K-contiguous (pass the
 * 1x1 weight transposed, [Cin][Cmid]); K, lda, ldb multiples of 8. */
int peclr_gemm_add_bf16(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                        int ldc, const void* addend, int ldd, peclr_stream_t stream);
/* the same kernel for IEEE fp16 activations (precision=16 / native AMP): v_mfma_f32_32x32x16_f16 */
int peclr_gemm_add_f16(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C,
                        int ldc, const void* addend, int ldd, peclr_stream_t stream);
/* fp32 GEMM on the bf16 matrix cores, fp32 accuracy: C[M,N] = A[M,K] . B[N,K]^T (+ addend[M,N], nullable), all fp32,
 * both operands K-contiguous.  Every fp32 operand is split EXACTLY into three bf16 numbers (x = h + m + l by
 * truncation); six of the nine partial products run on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, the three
 * dropped ones are <= 2^-24 of the product (one fp32 rounding).  2.67x the fp32 MFMA rate.  Serves the GEMM-shaped fp32
 * work of the backbone (same call sites as peclr_gemm_add_f32: the 1x1-convolution input gradient + residual gradient;
 * and the 1x1 convolutions of the torchvision Bottleneck, resnet_model.py:15).  K, lda, ldb multiples of 4.       */
int peclr_gemm_x6_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                      const float* addend, int ldd, peclr_stream_t stream);
/* The 1x1 weight gradient on the same scheme: dW[M=Cout, N=Cin] = A[K=R, M]^T . B[K=R, N] with A = dY and B = X as they
 * lie in NHWC memory (K is the slow dimension of both; lda >= M, ldb >= N; M, N, lda, ldb multiples of 4).  K is split
 * over peclr_gemm_x6_tn_slabs(M, N, K) workgroup rows, each writing one fp32 slab [M][N]; add them with
 * peclr_slab_reduce_f32 (fixed order: the result is deterministic, unlike the atomically accumulated split-K weight
 * gradients of the library kernels).  Replaces MIOpen's fp32 weight gradient of the 1x1 convolutions.            */
int peclr_gemm_x6_tn_slabs(int M, int N, int K);
int peclr_gemm_x6_tn_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* slabs,
                         int n_slabs, peclr_stream_t stream);
/* Optional fusion of a BatchNorm2d(+ReLU) layer's backward REDUCTION into the GEMM that produces the gradient arriving at
 * that layer (the input gradient of the convolution that consumed the layer's output): host struct of device pointers.
 * The epilogue sums, per row block and column, dY' and dY' * xhat (dY' = dY where the ReLU passed -- recomputed from the
 * layer's input x with scale / shift, or read from its 1-bit mask; xhat = (x - mean) * invstd) into
 * partial[ceil(M / tile_rows)][2][N], the layout peclr_bn2d_bwd_finalize_f32 combines with n_split = ceil(M / tile_rows):
 * the separate peclr_bn2d_bwd_reduce pass over dY and x is not needed.  Fixed summation order. */
typedef struct {
    const float* x;             /* the layer's input [M][N], contiguous */
    const float* mean;          /* [N] */
    const float* invstd;        /* [N] */
    const float* scale_shift;   /* [2][N] of the forward */
    const uint32_t* relu_mask;  /* [M][N / 32] or NULL (mask recomputed from x) */
    int relu;
    float* partial;
} peclr_bn_bwd_fuse;

/* "Pair" arithmetic of the packed-weight GEMMs below (round 6; every entry point that takes `pair`: NULL = the six-product
 * arithmetic on planes of peclr_x6_pack_f32).  Non-NULL: Bp holds planes of peclr_x6_pack_pair_f32 -- the weight multiplied by a
 * power of two (*w_scale) and split into TWO fp16 numbers hi + lo -- and the activation operand is scaled and split the same way in
 * the kernel by the power of two that puts *a_absmax (= max |A| over the whole tensor; written by the pass that produced A:
 * `absmax_out` of the peclr_bn2d_* entry points) into [2^14, 2^15).  Three products (hi.hi, hi.lo, lo.hi) on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation instead of six on the bf16 instruction; the accumulators are multiplied by
 * 1 / (s_a s_w) (exact).  22 - 23 of fp32's 24 significand bits per operand: error against float64 in the class of an fp32 BLAS
 * GEMM (measured per layer on real tensors: tools/exp/fp16_pair_probe.py, tools/exp/pair_probe.py).
 * Both pointers are DEVICE pointers to one float, read by the kernel: no host synchronisation.                                 */
typedef struct {
    const float* a_absmax;
    const float* w_scale;
} peclr_x6_pair;

/* Second generation of the same scheme for a WEIGHT operand (a parameter: constant for a whole step, used by forward,
 * input gradient and fused entry gradient of a 1x1 convolution, resnet_model.py:15): peclr_x6_pack_f32 splits the
 * weight ONCE into three bf16 planes in MFMA fragment order (per 128 output columns x 16 k one 12 KiB chunk of twelve
 * 1 KiB pieces [32-column block][plane]); peclr_gemm_x6p_f32 streams those chunks into LDS by LDS-DMA and only splits the
 * activation operand in the kernel (once per 128 output columns, by the one wave that owns the row).
 *   desc_table: DEVICE array of `count` entries of 8 int64 {src fp32 matrix, dst planes, n, k, ld (floats), transposed,
 *   first chunk, 0}; B_t[n][k] = src[n * ld + k] (transposed = 0: forward, W[Cout][Cin]) or src[k * ld + n]
 *   (transposed = 1: input gradients, B_t = W^T) or, for a T-tap filter W[Cout][T][Cin] read for its input gradient,
 *   transposed = T: B_t[ci][tap * Cout + co] = src[(co * T + tap) * ld + ci]; n % 128 == 0, k % 16 == 0; chunks per matrix = (n / 128) * (k / 16);
 *   dst holds peclr_x6_pack_bytes(n, k) = 6 n k bytes.  One launch packs every matrix of the table.
 *   peclr_gemm_x6p_f32: C[M,N] = A[M,K] . B_t^T (+ addend); tile_rows 256, 128 or 0 (= peclr_gemm_x6p_tile_rows).
 *   stat_partial (nullable; needs stat_shift[N]): the training-mode BatchNorm2d statistics of C (the `conv -> bn` pair of
 *   the torchvision Bottleneck, resnet_model.py:15) from the accumulators, so that no separate pass re-reads the tensor
 *   the GEMM just wrote: float [ceil(M / tile_rows)][2][N] per-row-block sums of (C - shift) and (C - shift)^2, then one
 *   row [N] holding the shift -- exactly the `partial` layout of peclr_bn2d_stats, so peclr_bn2d_finalize_f32 (or the
 *   synchronised route's peclr_bn2d_combine_f64) takes it with n_split = ceil(M / tile_rows).  Fixed summation order. */
int64_t peclr_x6_pack_bytes(int N, int K);
int peclr_x6_pack_f32(const void* desc_table, int count, int total_chunks, peclr_stream_t stream);
/* ... and as fp16 pairs (see peclr_x6_pair): same descriptor table, dst of peclr_x6_pack_pair_bytes(n, k) = 4 n k bytes (8 KiB
 * chunks of eight pieces [32-column block][hi | lo]).  Two launches over the same grid: peclr_x6_absmax_f32 leaves max |W| of
 * matrix d in absmax[d] (float [count]); peclr_x6_pack_pair_f32 multiplies matrix d by the power of two that maximum gives,
 * splits, and leaves the power of two in scales[d] (float [count]: what the GEMMs take as peclr_x6_pair.w_scale).              */
int64_t peclr_x6_pack_pair_bytes(int N, int K);
int peclr_x6_absmax_f32(const void* desc_table, int count, int total_chunks, float* absmax, peclr_stream_t stream);
int peclr_x6_pack_pair_f32(const void* desc_table, int count, int total_chunks, const float* absmax, float* scales,
                           peclr_stream_t stream);
int peclr_gemm_x6p_tile_rows(int M, int N, int K);
int peclr_gemm_x6p_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                       const float* addend, int ldd, int tile_rows, const float* stat_shift, float* stat_partial,
                       const peclr_bn_bwd_fuse* bn_bwd, const peclr_x6_pair* pair, peclr_stream_t stream);
/* The same product when the addend is the COMPACT input gradient of a 1x1 / stride-2 convolution: the M rows are the pixels
 * of H x W images (H, W even, M a multiple of H * W), addend_half = [images][H / 2][W / 2][ldd], and only the rows at even
 * (h, w) add addend_half[(h / 2, w / 2)].  This is the entry gradient of a ResNet layer's first block -- dY1 . W1 (main
 * branch) + the shortcut's transposed convolution -- without the 4x larger, three-quarters-zero tensor MIOpen's strided
 * input gradient writes and the addend pass reads (resnet_model.py:15; torchvision Bottleneck.downsample).               */
int peclr_gemm_x6p_s2add_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                             const float* addend_half, int ldd, int H, int W, int tile_rows, const peclr_bn_bwd_fuse* bn_bwd,
                             const peclr_x6_pair* pair, peclr_stream_t stream);
/* ... and when the addend is a gradient that still has to pass a ReLU: addend_mask = the 1-bit mask peclr_bn2d_apply wrote
 * for that ReLU ([M][N / 32] words, bit c % 32 of word c / 32 = output c was positive); addend elements whose bit is clear
 * count as zero.  This is the entry gradient of a residual block whose shortcut is the identity: dY1 . W1 +
 * relu'(out) * d(out) -- the second term is what the block's last BatchNorm backward would otherwise write out as the
 * residual's gradient (4 B per element written, then read back here).  N % 32 == 0.                                       */
int peclr_gemm_x6p_maskadd_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                               const float* addend, int ldd, const unsigned* addend_mask, int tile_rows,
                               const peclr_bn_bwd_fuse* bn_bwd, const peclr_x6_pair* pair, peclr_stream_t stream);
/* 3x3 / stride-1 / padding-1 convolution of an NHWC fp32 tensor (the middle convolution of the torchvision Bottleneck,
 * resnet_model.py:15) as an implicit GEMM on the same kernel: rows = output pixels, K = 9 * Cin ordered (tap, channel); the
 * activation rows of a k-step come from the pixel the tap points at (zeros outside the image: `zeros` = >= 64 bytes of
 * zeros), the weights from planes packed out of W[Cout][3][3][Cin] (channels_last storage) seen as [Cout][9 * Cin].
 * flip = 1 computes the INPUT GRADIENT dX = conv(dY, flipped filter): pass dY as X, Cin := Cout, Cout := Cin and planes
 * packed with `transposed` = 9 (B_t[ci][tap * Cout + co] = W[co][tap][ci]).  addend, tile_rows and the BatchNorm
 * statistics outputs as in peclr_gemm_x6p_f32.  Cin % 16 == 0, Cout % 128 == 0.                                      */
int peclr_conv3x3_x6p_f32(int NB, int H, int W, int Cin, int Cout, const float* X, const void* Bp, float* Y,
                          const float* addend, int flip, int tile_rows, int variant, const float* zeros, const float* stat_shift,
                          float* stat_partial, const peclr_bn_bwd_fuse* bn_bwd, const peclr_x6_pair* pair, peclr_stream_t stream);
/* (variant 0: every wave loads and splits its rows once per tap; 1: per 16-channel chunk the workgroup splits the pixels its
 *  nine taps touch once into shared planes and the taps read their fragments at the tap's offset -- W <= 64, else as 0.)  */
/* Weight gradients, second generation: C[M, taps * N] = sum_k A[k, M] . B[k shifted by the tap, N], the contraction over the
 * ROWS of two NHWC activations (A = dY [R, Cout], B = X [R, Cin]).  taps = 1: dW = dY^T X of a 1x1 convolution.  taps = 9:
 * the nine [Cout, Cin] products of a 3x3 / stride-1 / padding-1 convolution's weight gradient, X read at the pixel each
 * tap points at (H x W images, K = images * H * W rows; zeros outside the image: `zeros` = >= 64 bytes of zeros), written
 * as columns tap * N + n -- the [Cout][3][3][Cin] storage of a channels_last weight.  K is split over
 * peclr_gemm_x6t_slabs(M, N, K, taps) workgroup rows, each writing one fp32 slab [M][taps * N]; peclr_slab_reduce_f32 adds
 * them in a fixed order (deterministic).  256 x 256 output tiles where the problem has them (an operand row is split once
 * per 256 columns of the other operand), k-step 16 with double-buffered planes; 64-wide sides (layer1) get 64 x 256 /
 * 256 x 64 / 64 x 128 tiles and, for taps = 9, a 64 x 64 block with the taps in two halves.  stride = 2 (the first block of
 * layers 2-4): A's K rows are the H x W OUTPUT pixels, B's 4 K rows the pixels of the 2H x 2W inputs, read at
 * (2 oh + dh, 2 ow + dw) (taps = 1: the 1x1 shortcut, dh = dw = 0; taps = 9: padding 1).  M, N, lda, ldb multiples of 4;
 * W >= 6 when taps = 9 or stride = 2.  Replaces MIOpen's fp32 weight gradients of the Bottleneck's convolutions
 * (resnet_model.py:15).                                                                                                  */
int peclr_gemm_x6t_slabs(int M, int N, int K, int taps);
int peclr_gemm_x6t_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* slabs, int n_slabs,
                       int taps, int H, int W, int stride, const float* zeros, peclr_stream_t stream);
/* Input gradient of the 3x3 / padding-1 / STRIDE-2 convolution of a layer's first block (resnet_model.py:15), on the same
 * kernel: dY [NB, Ho, Wo, Cout] NHWC -> dX [NB, 2 Ho, 2 Wo, Cin].  Input pixel (2 i + ph, 2 j + pw) receives filter row a only
 * where ph + 1 - a is even, so the transposed convolution splits into four dense ones, one per parity class (ph, pw), with
 * 1, 2, 2 and 4 taps (gridDim.y = class): the same 9 Cout multiply-adds per OUTPUT pixel as the forward, none against the
 * zeros a dilated gradient would hold.  Bp: planes of the filter packed for the input gradient (peclr_x6_pack_f32,
 * transposed = 9, as for peclr_conv3x3_x6p_f32 with flip).  bn_bwd: dX is the gradient arriving at that BatchNorm layer
 * (partial: 4 * ceil(NB Ho Wo / tile_rows) row blocks, class by class).  Cin % 64 == 0, Cout % 16 == 0.  Replaces MIOpen's
 * igemm_bwd on these three convolutions.                                                                                  */
int peclr_conv3x3_s2_dgrad_x6p_f32(int NB, int Ho, int Wo, int Cout, int Cin, const float* dY, const void* Bp, float* dX,
                                   int tile_rows, const float* zeros, const peclr_bn_bwd_fuse* bn_bwd, const peclr_x6_pair* pair,
                                   peclr_stream_t stream);
/* Forward of the STRIDE-2 convolutions of a ResNet layer's first block on the same kernel: taps = 9 the 3x3 / padding-1
 * convolution, taps = 1 the 1x1 downsample convolution; X [NB, H, W, Cin] NHWC (H, W even), Y [NB, H/2, W/2, Cout]; output pixel
 * (oh, ow) reads input pixel (2 oh + dh, 2 ow + dw).  Planes packed as for the stride-1 forward; optional BatchNorm
 * statistics of Y.  (Weight gradients: peclr_gemm_x6t_f32 with stride = 2; input gradients: above / peclr_gemm_x6p_s2add_f32.)  */
int peclr_conv_s2_x6p_f32(int NB, int H, int W, int Cin, int Cout, int taps, const float* X, const void* Bp, float* Y,
                          int tile_rows, const float* zeros, const float* stat_shift, float* stat_partial,
                          const peclr_x6_pair* pair, peclr_stream_t stream);
int peclr_gemm_pick_split_k(int M, int N, int K);

/* out[i] = sum_s slabs[s][i] (+ bias[i % cols] if non-null), i < rows*cols. */
int peclr_slab_reduce_f32(const float* slabs, int n_slabs, int rows, int cols, const float* bias,
                          float* out, peclr_stream_t stream);

/* ---- K2: BatchNorm1d (train or eval) + ReLU --------------------------------------------
 * Replaces nn.BatchNorm1d(H) + nn.ReLU (simclr_model.py:27-28).  Input = split-K slabs of
 * the first Linear plus its bias.  training != 0: batch statistics over all M rows (biased
 * variance), running stats updated with momentum and the unbiased variance,
 * num_batches_tracked incremented (all three nullable).  training == 0: running stats.
 * Outputs: a_pre [M,H] (pre-BN activations, kept for the backward), a_out [M,H],
 * save_mean[H], save_invstd[H].                                                           */
int peclr_bn_relu_fwd_f32(const float* a_slabs, int n_slabs, const float* bias, int M, int H,
                          const float* gamma, const float* beta, float eps, float momentum,
                          int training, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float* a_pre, float* a_out,
                          float* save_mean, float* save_invstd, peclr_stream_t stream);
/* Backward of the above: d_a_out -> d_a_pre [M,H], dgamma, dbeta, dbias[H].  training != 0:
 * batch statistics took part in the forward (full BN backward); training == 0: the statistics
 * were constants (frozen BN).                                                               */
int peclr_bn_relu_bwd_f32(const float* d_a_out, const float* a_pre, const float* save_mean,
                          const float* save_invstd, const float* gamma, const float* beta, int M,
                          int H, int training, float* d_a_pre, float* dgamma, float* dbeta,
                          float* dbias, peclr_stream_t stream);

/* ---- K3..K7: projection stats + normalise + un-translate + un-rotate + normalise -------
 * Replaces Hybrid2Model.get_projection_stats / F.normalize / translate_encodings /
 * rotate_encoding / F.normalize (hybrid2_model.py:40-85, utils.py:271-346).
 * p_slabs: [n_slabs][M][D] split-K partials of the second Linear.  D must be 128.
 * jitter_*: int64 [n_pairs] per view exactly as the batch dict holds them
 * (data_set.py:357-384); extent_x / extent_y are float(image_shape[0]) / float(shape[1]).
 * angle1/2: float64 [n_pairs] per view, NOT negated (the kernel applies the minus signs of
 * hybrid2_model.py:74,80).  Pointers of disabled flags may be null.
 * Outputs: p_out [M,D] (reduced raw projections), z_out [M,D], norms [2][M] (clamped
 * ||p|| and clamped post-alignment norm), row_stats [M][8] per-sample
 * {x_mean,x_median,x_min,x_max,y_mean,y_median,y_min,y_max} (nullable).                   */
int peclr_align_fwd_f32(const float* p_slabs, int n_slabs, int M, int D, int n_pairs, int flags,
                        const int64_t* jitter_x1, const int64_t* jitter_x2,
                        const int64_t* jitter_y1, const int64_t* jitter_y2, float extent_x,
                        float extent_y, const double* angle1, const double* angle2, float* p_out,
                        float* z_out, float* norms, float* row_stats, peclr_stream_t stream);
/* Backward: dz [M,D] -> dp [M,D].  Range and centroid are constants (utils.py:312,338-339). */
int peclr_align_bwd_f32(const float* dz, const float* p, const float* z, const float* norms, int M,
                        int D, int n_pairs, int flags, const double* angle1, const double* angle2,
                        float* dp, peclr_stream_t stream);

/* ---- K8: NT-Xent ------------------------------------------------------------------------
 * Replaces vanila_contrastive_loss (utils.py:154-186) without materialising S, exp(S), the
 * mask or the M x (M-1) gather.  One entry point per kernel launch.
 * z_rows: the Mr rows this call owns (global row index = row_offset + r); z_all: all Mg rows
 * (negatives).  Single GPU: z_rows == z_all, Mr == Mg, row_offset == 0.
 *
 * peclr_ntxent_jsplit: the library's column-split policy (how many key-column slices the
 * grid is split into) for the forward (backward == 0) or backward (backward != 0) launch;
 * callers size `partial` / `dz_slabs` with it.
 * peclr_ntxent_fwd_f32: partial[jsplit][Mr] = per-slice sums of exp(S_ij/tau) over j != i,
 * pos[Mr] = S_{i,partner(i)}/tau; sim_out (nullable): [Mr][Mg] per-pair similarities.
 * peclr_ntxent_finalize_f32 (one workgroup, fixed summation order): row_lse[Mr] =
 * log sum_s partial[s][r]; out17[16] = loss_scale * sum_r (row_lse[r] - pos[r]); when row_stats
 * is non-null, out17[0..15] = the batch-mean projection statistics (hybrid2_model.py:92-106)
 * of the [2*n_pairs_stats][8] row_stats matrix.
 * peclr_ntxent_bwd_f32: dz_slabs[s][Mr][D] = per-slice partials of
 *   (*dloss) * grad_scale * inv_tau * sum_{j != i} [e_ij (1/neg_i + 1/neg_j) - 2 [j == partner(i)]] z_j
 * (sum the slabs with peclr_slab_reduce_f32; jsplit == 1 writes the dense dz directly).
 * lse_all[Mg] = row_lse of every global row (all-gathered on multi-GPU); grad_scale = 1/Mg for
 * the reference's mean reduction; dloss: device scalar (the incoming gradient of the loss).  */
int peclr_ntxent_jsplit(int Mr, int Mg, int backward);
int peclr_ntxent_fwd_f32(const float* z_rows, int Mr, int row_offset, const float* z_all, int Mg,
                         int D, int n_half, float inv_tau, float* sim_out, float* partial,
                         float* pos, int jsplit, peclr_stream_t stream);
int peclr_ntxent_finalize_f32(const float* partial, int jsplit, const float* pos, int Mr,
                              float loss_scale, float* row_lse, const float* row_stats,
                              int n_pairs_stats, float* out17, peclr_stream_t stream);
int peclr_ntxent_bwd_f32(const float* z_rows, int Mr, int row_offset, const float* z_all, int Mg,
                         int D, int n_half, float inv_tau, const float* lse_all,
                         const float* dloss, float grad_scale, float* dz_slabs, int jsplit,
                         peclr_stream_t stream);

/* ---- optimiser: LARSWrapper(Adam) fused over a list of tensors --------------------------
 * Replaces pl_bolts LARSWrapper.step + torch.optim.Adam.step as wired by
 * BaseModel.configure_optimizers (base_model.py:57-104).  Two launches per parameter group
 * instead of ~15 launches + 2 host syncs per tensor.
 * ptrs: device array [4][n_tensors] of float* {param, grad, exp_avg, exp_avg_sq};
 * sizes: device array [n_tensors] of int64 element counts.  The work list is n_chunks chunks
 * of at most PECLR_OPT_CHUNK elements: chunk_tensor[c] / chunk_offset[c] give the tensor and
 * the element offset of chunk c (chunks of one tensor are consecutive), and
 * tensor_chunk_begin[t] .. tensor_chunk_begin[t+1] is tensor t's chunk range.
 * peclr_lars_sumsq_f32: norms_ws[2][n_chunks] = per-chunk sums of squares of param and grad.
 * peclr_lars_adam_update_f32: ONE launch for all parameter groups (the reference has two: decayed /
 * not decayed, base_model.py:30-51): tensor_group[t] (device, nullable when n_groups == 1) selects
 * the tensor's entry of the HOST arrays group_lr / group_weight_decay (n_groups <= 8, copied into
 * the kernel arguments).  It combines a tensor's chunk sums in a fixed order (bit-reproducible),
 * applies the LARS trust ratio + weight decay and the Adam update.
 * device_hyper (nullable): DEVICE float[18] = {lr[8], weight_decay[8], bias_corr1, bias_corr2}; when
 * given it overrides the by-value scalars, so a hipGraph that captured the launch can be replayed
 * with new per-step values.
 * use_lars == 0: plain Adam with L2 weight decay (torch.optim.Adam semantics; norms_ws unused).
 * use_lars == 2: LARS, and the scaled gradient (g + wd*p)*trust is ALSO written back over grad, as
 * pl_bolts' LARSWrapper.update_p leaves it (it mutates p.grad in place before the wrapped Adam
 * step); with use_lars == 1 it is consumed in registers only (4 B/param less traffic).
 * bias_corr1/2 = 1 - beta^step, computed by the host.                                        */
#define PECLR_OPT_CHUNK 4096
int peclr_lars_sumsq_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                         const int32_t* chunk_tensor, const int64_t* chunk_offset, int n_chunks,
                         float* norms_ws, peclr_stream_t stream);
int peclr_lars_adam_update_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                               const int32_t* chunk_tensor, const int64_t* chunk_offset,
                               const int32_t* tensor_chunk_begin, const int32_t* tensor_group,
                               int n_chunks, const float* norms_ws, const float* device_hyper,
                               const float* group_lr, const float* group_weight_decay, int n_groups,
                               float beta1,
                               float beta2, float adam_eps, float bias_corr1, float bias_corr2,
                               int use_lars, float lars_eta, float lars_eps, int lars_clip,
                               peclr_stream_t stream);

/* ---- precision=16: the dynamic loss scaler folded into the optimiser step -------------------
 * Replaces torch.cuda.amp.GradScaler.unscale_/step/update around the optimiser, which Lightning 1.0.8's
 * native-AMP plugin runs when the reference trains at its default `precision: 16`
 * (training_config.json:9, peclr_training.py:78-79).  GradScaler.step reads found_inf back to the host to
 * decide whether to step; here the decision is taken on the device, so the step can live in a hipGraph:
 *   peclr_lars_sumsq_amp_f32        the gradients hold scale*g: norms of g = grad/scale, and
 *                                   amp->found_inf = 1 if any gradient element is inf / nan;
 *   peclr_lars_adam_update_amp_f32  returns without touching param / moments when found_inf is set (with
 *                                   use_lars == 2 it leaves grad/scale in grad, as unscale_ does); else the
 *                                   update on grad/scale with bias corrections 1 - beta^(good_steps + 1), evaluated in
 *                                   double from the double betas (bit-equal to the host's `1 - beta ** step`);
 *   peclr_amp_update                GradScaler.update: found_inf ? scale *= backoff, tracker = 0
 *                                   : (good_steps += 1; ++tracker == interval ? scale *= growth, tracker = 0),
 *                                   then found_inf = 0.  One thread.
 * Launch the three in this order on one stream.  `amp` is DEVICE memory, 16 bytes, owned by the caller
 * (initialise scale = 65536, the rest 0 for torch's defaults).                                         */
typedef struct peclr_amp_state {
    float scale;          /* current loss scale                                   */
    float found_inf;      /* 0 / 1, raised by the sumsq pass, cleared by update   */
    int32_t growth_tracker; /* consecutive steps without inf / nan                */
    int32_t good_steps;   /* optimiser steps actually taken (Adam's `step`)       */
} peclr_amp_state;
int peclr_lars_sumsq_amp_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                             const int32_t* chunk_tensor, const int64_t* chunk_offset, int n_chunks,
                             float* norms_ws, peclr_amp_state* amp, peclr_stream_t stream);
int peclr_lars_adam_update_amp_f32(float* const* ptrs, const int64_t* sizes, int n_tensors,
                                   const int32_t* chunk_tensor, const int64_t* chunk_offset,
                                   const int32_t* tensor_chunk_begin, const int32_t* tensor_group,
                                   int n_chunks, const float* norms_ws, const float* device_hyper,
                                   const float* group_lr, const float* group_weight_decay, int n_groups,
                                   double beta1, double beta2, float adam_eps, int use_lars, float lars_eta,
                                   float lars_eps, int lars_clip, const peclr_amp_state* amp,
                                   peclr_stream_t stream);
int peclr_amp_update(peclr_amp_state* amp, float growth_factor, float backoff_factor, int growth_interval,
                     peclr_stream_t stream);

/* ---- backbone glue: fused BatchNorm2d (+ residual add) (+ ReLU), NHWC --------------------
 * Replaces nn.BatchNorm2d + `out += identity` + nn.ReLU between the convolutions of the
 * torchvision ResNet blocks the reference builds (resnet_model.py:15, norm_layer=nn.BatchNorm2d).
 * Activations are NHWC (torch.channels_last): x, residual, y, dy, dx, d_residual are row-major
 * [R = N*H*W][C] of io_dtype (PECLR_DTYPE_F32, or PECLR_DTYPE_BF16 / PECLR_DTYPE_F16 for autocast backbones);
 * statistics, parameters and arithmetic are fp32.  With W = 4 (fp32) / 8 (bf16) channels per
 * 16-byte word, C/W must divide 256 or be a multiple of it (every ResNet width).
 * One entry point = one launch:
 *   forward (training): peclr_bn2d_stats -> peclr_bn2d_finalize_f32 -> peclr_bn2d_apply
 *   forward (eval)    :                     peclr_bn2d_finalize_f32 -> peclr_bn2d_apply
 *   backward          : peclr_bn2d_bwd_reduce -> peclr_bn2d_bwd_finalize_f32 -> peclr_bn2d_bwd_apply
 * partial (forward): [n_split*2 + 1][C] floats = row-slice partial sums (of x - shift and its
 * square) + one row holding the shift; shift (nullable): [C] vector to subtract, default = row 0
 * of x.  partial (backward): [n_split*2][C].  n_split from peclr_bn2d_n_split, or any >= 1.
 * Synchronised statistics across data-parallel ranks (optional; every rank passes the SAME shift,
 * e.g. running_mean): peclr_bn2d_stats -> peclr_bn2d_combine_f64 (totals: double [2*C+1]; the
 * caller stores its row count in the last element) -> the caller SUM-all-reduces totals -> peclr_bn2d_finalize_totals_f32; backward:
 * peclr_bn2d_bwd_reduce -> peclr_bn2d_combine_f64 -> all-reduce ->
 * peclr_bn2d_bwd_finalize_totals_f32 (dgamma/dbeta from the LOCAL totals, dx coefficients from the
 * GLOBAL ones).
 * scale_shift: [2][C] = {gamma*invstd, beta - mean*gamma*invstd}; coef: [2][C] scratch for dx.
 * ReLU mask in the backward: recomputed from x when neither y nor relu_mask is given (valid only
 * if no residual was added); read from relu_mask ([R][C/32] uint32, 1 bit per element, written by
 * peclr_bn2d_apply when its relu_mask argument is non-null; needs C % 32 == 0) or, failing that,
 * from the forward output y.  d_residual (nullable) receives the masked dy.
 * peclr_bn2d_finalize_f32 / peclr_bn2d_bwd_finalize_f32 REWRITE long partial tables (hence no const): with n_split >= 2048 the
 * rows are first summed in slices of 512 by a launch of their own that leaves each slice's sum in the slice's first row and
 * zeros in its other rows -- the table keeps its totals, so a second finalize / combine of the same table gives the same
 * result, but the individual row-block partials are gone.                                      */
#define PECLR_DTYPE_F32 0
#define PECLR_DTYPE_BF16 1
#define PECLR_DTYPE_F16 2   /* IEEE half: precision=16 (native AMP), the reference's default */
int peclr_bn2d_n_split(int R, int C, int io_dtype);
int peclr_bn2d_stats(const void* x, int io_dtype, int R, int C, const float* shift,
                     float* partial, int n_split, peclr_stream_t stream);
int peclr_bn2d_finalize_f32(float* partial, int n_split, int R, int C, int training,
                            float eps, float momentum, const float* gamma, const float* beta,
                            float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* save_mean, float* save_invstd,
                            float* scale_shift, peclr_stream_t stream);
int peclr_bn2d_combine_f64(const float* partial, int n_split, int C, double* totals,
                           peclr_stream_t stream);
int peclr_bn2d_finalize_totals_f32(const double* totals, const float* shift, int C,
                                   float eps, float momentum, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var,
                                   int64_t* num_batches_tracked, float* save_mean,
                                   float* save_invstd, float* scale_shift, peclr_stream_t stream);
int peclr_bn2d_bwd_finalize_totals_f32(const double* local_totals, const double* global_totals,
                                       int C, int training, const float* scale_shift,
                                       float* dgamma, float* dbeta, float* coef,
                                       peclr_stream_t stream);
/* absmax_out (nullable, here and in the other passes that take it): the pass also leaves max |value it wrote| in *absmax_out -- one
 * atomic maximum per wave on a float its caller ZEROED beforehand (non-negative floats order like their bit patterns; a maximum
 * does not depend on the order, so the result is deterministic).  It costs the pass nothing measurable (it is HBM-bound) and is
 * what the "pair" GEMMs that consume the tensor take as peclr_x6_pair.a_absmax: every activation and every gradient a residual
 * block's convolution reads is written by one of these passes.                                                                  */
int peclr_bn2d_apply(const void* x, const void* residual, int io_dtype, int R, int C,
                     const float* scale_shift, int relu, void* y, uint32_t* relu_mask, float* absmax_out,
                     peclr_stream_t stream);
/* The last pass of a layer's FIRST block, whose shortcut is conv1x1 -> BatchNorm2d (torchvision Bottleneck.downsample behind
 * resnet_model.py:15): y = (relu)(bn(x) + bn_s(res_x)).  res_x is the INPUT of the shortcut's BatchNorm, res_scale_shift its
 * [2][C] table; the residual is fmaf(res_x, scale, shift) rounded to the storage format -- the value peclr_bn2d_apply would
 * have written and this pass read back, so the result is bit-identical -- and the shortcut's apply pass and output tensor
 * disappear.  relu_mask as in peclr_bn2d_apply.                                                                            */
int peclr_bn2d_apply_res_bn(const void* x, const void* res_x, const float* res_scale_shift, int io_dtype, int R, int C,
                            const float* scale_shift, int relu, void* y, uint32_t* relu_mask, float* absmax_out,
                            peclr_stream_t stream);
int peclr_bn2d_bwd_reduce(const void* dy, const void* x, const void* y, const uint32_t* relu_mask,
                          int io_dtype, int R, int C, int relu, const float* save_mean,
                          const float* save_invstd, const float* scale_shift, float* partial,
                          int n_split, peclr_stream_t stream);
int peclr_bn2d_bwd_finalize_f32(float* partial, int n_split, int R, int C, int training,
                                const float* scale_shift, float* dgamma, float* dbeta,
                                float* coef, peclr_stream_t stream);
int peclr_bn2d_bwd_apply(const void* dy, const void* x, const void* y, const uint32_t* relu_mask,
                         int io_dtype, int R, int C, int relu, const float* save_mean,
                         const float* save_invstd, const float* scale_shift, const float* coef,
                         void* dx, void* d_residual, float* absmax_out, peclr_stream_t stream);

/* Encoder tail: the last residual block's BatchNorm2d + `out += identity` + ReLU, then
 * AdaptiveAvgPool2d((1,1)) and `.flatten(1)` (features[7][-1].bn*, features[8] and the flatten of
 * ResNetModel.forward, resnet_model.py:24-26,47), as one pass that never writes the [N][HW][C]
 * activation: only `pooled` = its per-image channel means, fp32 [N][C] -- the encoder output the
 * projection head's first GEMM (peclr_gemm_f32, K1) reads directly -- and the 1-bit ReLU mask.
 * Forward: peclr_bn2d_stats -> peclr_bn2d_finalize_f32 (on x, R = N*HW) -> peclr_bn2d_apply_avgpool.
 * Backward: the pool's gradient is a broadcast, dy[n][p][c] = d_pooled[n][c] / HW, formed on the fly
 * (no [R][C] gradient tensor): peclr_bn2d_bwd_reduce_avgpool -> peclr_bn2d_bwd_finalize_f32 ->
 * peclr_bn2d_bwd_apply_avgpool (dx and d_residual = masked dy, both [R][C] of io_dtype).
 * Needs C % 32 == 0 and a residual (every ResNet's last block).                                  */
int peclr_bn2d_apply_avgpool(const void* x, const void* residual, int io_dtype, int N, int HW, int C,
                             const float* scale_shift, float* pooled, uint32_t* relu_mask,
                             peclr_stream_t stream);
int peclr_bn2d_bwd_reduce_avgpool(const float* d_pooled, const void* x, const uint32_t* relu_mask,
                                  int io_dtype, int N, int HW, int C, const float* save_mean,
                                  const float* save_invstd, const float* scale_shift, float* partial,
                                  int n_split, peclr_stream_t stream);
int peclr_bn2d_bwd_apply_avgpool(const float* d_pooled, const void* x, const uint32_t* relu_mask,
                                 int io_dtype, int N, int HW, int C, const float* save_mean,
                                 const float* save_invstd, const float* scale_shift, const float* coef,
                                 void* dx, void* d_residual, float* absmax_out, peclr_stream_t stream);

/* Stem: BatchNorm2d + ReLU + MaxPool2d(3, stride 2, padding 1) in one pass (features[1..3] of the
 * torchvision ResNet the reference wraps, resnet_model.py:15-26); x is [N][H][W][C] NHWC, the pooled
 * output [N][PH][PW][C] with PH = (H-1)/2 + 1.  The un-pooled activation is never written.  Forward:
 * peclr_bn2d_stats -> peclr_bn2d_finalize_f32 (as above, on x) -> peclr_bn2d_pool_apply, which also
 * writes, per pooled element, a 1-byte tap code (3*dh+dw of the first maximum, row-major, torch's
 * rule) and the x value at that maximum (x_at_max, same dtype/shape as y).  Backward:
 * peclr_bn2d_pool_bwd_reduce (a streaming reduction over dy and x_at_max; partial [n_split][2][C],
 * n_split from peclr_bn2d_pool_n_split or any >= 1) -> peclr_bn2d_bwd_finalize_f32 with R = N*H*W
 * -> peclr_bn2d_pool_bwd_apply (dx on the un-pooled grid: every element gathers from the <= 4
 * windows that hold it).  C/4 (fp32) or C/8 (bf16) must divide 256. */
int peclr_bn2d_pool_n_split(int N, int H, int W, int C, int io_dtype);
int peclr_bn2d_pool_apply(const void* x, int io_dtype, int N, int H, int W, int C,
                          const float* scale_shift, void* y, void* x_at_max, uint8_t* code, float* absmax_out,
                          peclr_stream_t stream);
int peclr_bn2d_pool_bwd_reduce(const void* dy_pool, const void* x_at_max, int io_dtype, int N, int H,
                               int W, int C, const float* save_mean, const float* save_invstd,
                               const float* scale_shift, float* partial, int n_split,
                               peclr_stream_t stream);
int peclr_bn2d_pool_bwd_apply(const void* dy_pool, const void* x, const uint8_t* code, int io_dtype,
                              int N, int H, int W, int C, const float* save_mean,
                              const float* save_invstd, const float* scale_shift, const float* coef,
                              void* dx, peclr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Two-view augmentation, pixel side.
 * Replaces, for a whole batch and both views at once, what SampleAugmenter.transform_sample does
 * per sample with OpenCV on CPU workers for the published recipe (reference
 * src/data_loader/sample_augmenter.py:47-129: rotate_sample :218-247 cv2.warpAffine, crop_sample
 * :166-186, resize_sample :188-216 cv2.resize INTER_AREA, color_jitter_sample :267-293 cv2 BGR<->HSV)
 * plus ToTensor + Normalize (src/data_loader/utils.py:283-293).  The parameter side (random draws,
 * crop box, rotation matrix) stays on the host: peclr_amd/augment.py.
 *
 * images : [B][H][W][3] uint8 (the reference's HWC images, one size per batch)
 * params : [n_views][B][PECLR_AUG_PARAM_DOUBLES] doubles per (view, sample):
 *            [0..5]  inverse (destination -> source) 2x3 affine of the rotation, row-major
 *            [6]     != 0: rotate            [7..10] crop window x0, y0, width, height (inside the image)
 *            [11]    != 0: colour jitter     [12..15] h, s, a, b factors
 * crops  : [n_views][B][H][W][3] uint8 scratch; only each window (at offset 0,0, row stride W) is written
 * out    : float32, logical shape [n_views*B][3][out_h][out_w]; channels_last != 0 stores it NHWC
 * mean/stdv : HOST pointers to 3 floats each (passed to the kernel by value).
 * 8-bit intermediates sit where the reference has them, so results are bit-identical to the
 * restatement in oracle/augment_oracle.py (whose pixel arithmetic is itself unpinned: OpenCV is
 * not available to check against). */
#define PECLR_AUG_PARAM_DOUBLES 16
int peclr_augment_warp_crop_u8(const uint8_t* images, int B, int H, int W, int n_views,
                               const double* params, uint8_t* crops, peclr_stream_t stream);
int peclr_augment_resize_color_norm(const uint8_t* crops, int B, int H, int W, int n_views,
                                    const double* params, int out_h, int out_w, const float* mean,
                                    const float* stdv, int channels_last, float* out,
                                    peclr_stream_t stream);

/* ---- 16-bit convolutions of the residual blocks (bf16 / fp16 autocast: BASELINE configs C3 / C5 and the reference's own
 * `precision: 16`, training_config.json:9, peclr_training.py:78-79), csrc/conv_h.hip.  They replace MIOpen's 16-bit
 * convolution kernels behind torchvision's Bottleneck / BasicBlock (resnet_model.py:15) -- forward and input gradient of the
 * 1x1 and 3x3 convolutions, stride 1 and 2 -- with fp32 accumulation on v_mfma_f32_32x32x16_{bf16,f16} and the same fused
 * epilogues as the fp32 kernels above: BatchNorm statistics of the output, the backward reduction of the BatchNorm layer the
 * output gradient arrives at, the residual branch's gradient as a dense / compact stride-2 / 1-bit-masked addend.
 * dtype = PECLR_DTYPE_BF16 or PECLR_DTYPE_F16 for activations, gradients and packed weights alike.
 *   peclr_h_pack: the weight operand, packed ONCE per optimiser step from the fp32 master weights (this is also the cast
 *   autocast performs per forward): device table of `count` entries of 8 int64 {src fp32 matrix, dst, n, k, ld (floats),
 *   transposed (0 / 1 / T as peclr_x6_pack_f32), first chunk, dtype}; per (128 columns, 32 k) one 8 KiB chunk of eight
 *   1 KiB pieces [32-column block][16-k half] in MFMA fragment order; n % 64 == 0, k % 32 == 0; dst holds
 *   peclr_h_pack_bytes(n, k) = 2 * roundup(n, 128) * k bytes.
 *   peclr_gemm_h: C[M,N] = A[M,K] . B_t^T (+ addend): add_h = add_w = 0 dense addend [M][ldd]; add_h, add_w > 0 the addend
 *   holds every second pixel of add_h x add_w images (compact stride-2 shortcut gradient); addend_mask: 1-bit ReLU mask of a
 *   dense addend ([M][N / 32]).  stat_shift / stat_partial and bn_bwd as in peclr_gemm_x6p_f32 (bn_bwd->x points at 16-bit
 *   rows).  Sums are taken of the ROUNDED 16-bit outputs.  lda, ldc, ldd multiples of 8.
 *   peclr_conv_h: taps = 9 (3x3, padding 1) or 1, stride 1 or 2, NHWC; flip = 1 (taps = 9, stride 1): the input gradient
 *   (X = dY, planes packed with transposed = 9).  `zeros`: >= 64 bytes of zeros.  Cin % 32 == 0, Cout % 64 == 0.
 *   tile_rows: 128 or 256 output rows per workgroup, 0 = the library's choice.  3x3 / stride 1 with rows of <= 62 pixels:
 *   PECLR_CONV_H_RING (and the choice of 0) runs the ring form -- a workgroup owns 256 consecutive pixels of the PADDED space
 *   NB x (H + 1) x (W + 1) (one shared zero row / column per image, as peclr_wgrad3_h) and fetches the 256 + 2 (W + 2) input
 *   pixels its nine taps read once per 32-channel chunk instead of once per tap.  peclr_conv_h_row_blocks: the rows the
 *   partial-sum tables (stat_partial: 2 * rows + 1, bn_bwd->partial: 2 * rows) of such a launch have (0: unsupported).
 *   peclr_conv3x3_s2_dgrad_h: input gradient of the 3x3 / stride-2 convolution by parity classes (as the fp32 entry point). */
#define PECLR_CONV_H_RING 1
int64_t peclr_h_pack_bytes(int N, int K);
int peclr_h_pack(const void* desc_table, int count, int total_chunks, peclr_stream_t stream);
int peclr_conv_h_tile_rows(int M, int N);
int peclr_conv_h_row_blocks(int NB, int H, int W, int Cout, int taps, int stride, int tile_rows);
int peclr_gemm_h(int dtype, int M, int N, int K, const void* A, int lda, const void* Bp, void* C, int ldc, const void* addend,
                 int ldd, int add_h, int add_w, const unsigned* addend_mask, int tile_rows, const float* stat_shift,
                 float* stat_partial, const peclr_bn_bwd_fuse* bn_bwd, peclr_stream_t stream);
int peclr_conv_h(int dtype, int NB, int H, int W, int Cin, int Cout, int taps, int stride, const void* X, const void* Bp, void* Y,
                 int flip, int tile_rows, const void* zeros, const float* stat_shift, float* stat_partial,
                 const peclr_bn_bwd_fuse* bn_bwd, peclr_stream_t stream);
int peclr_conv3x3_s2_dgrad_h(int dtype, int NB, int Ho, int Wo, int Cout, int Cin, const void* dY, const void* Bp, void* dX,
                             int tile_rows, const void* zeros, const peclr_bn_bwd_fuse* bn_bwd, peclr_stream_t stream);
/* Weight gradient of a 16-bit 1x1 convolution (csrc/wgrad_h.hip): dW[Cout][Cin] (fp32) = dY^T X over the rows of two NHWC
 * activations -- A = dY [K][lda] (M = Cout), B = X [K][ldb] (N = Cin); stride = 2 (the 1x1 / stride-2 shortcut): A's K rows are
 * the Ho x Wo output pixels, B holds the 2 Ho x 2 Wo input pixels.  Both operands go global -> LDS by LDS-DMA as they lie in
 * memory and are transposed by the LDS read (ds_read_b64_tr_b16).  K is split over peclr_wgrad_h_slabs(M, N, K) workgroup rows,
 * each writing one fp32 slab [M][N]; peclr_slab_reduce_f32 adds them in a fixed order (deterministic; fp32 gradient of the fp32
 * master weight).  Replaces MIOpen's 16-bit weight gradients (zero-fill + atomic split-K + cast) under resnet_model.py:15.
 * M, N multiples of 32, lda / ldb multiples of 8; `zeros`: >= 64 bytes of zeros. */
/* ... and of the 16-bit 3x3 / padding-1 / stride-1 convolutions: dW[Cout][3][3][Cin] (the storage of a channels_last weight) from
 * dY [images, H, W, M = Cout] and X [images, H, W, N = Cin].  The contraction runs over a PADDED linear pixel space (one shared
 * zero column per row, one shared zero row per image, applied by the addresses the LDS-DMAs fetch), in which tap (a, b) is a
 * uniform shift: X goes through a ring in LDS once and serves all nine taps by transposing reads at the taps' offsets.
 * M, N multiples of 64, W <= 62; slabs [peclr_wgrad3_h_slabs(...)][M][9 N] fp32, summed by peclr_slab_reduce_f32. */
int peclr_wgrad3_h_slabs(int M, int N, int images, int H, int W);
int peclr_wgrad3_h(int dtype, int M, int N, int images, int H, int W, const void* A, const void* B, float* slabs, int n_slabs,
                   const void* zeros, peclr_stream_t stream);
int peclr_wgrad_h_slabs(int M, int N, int K);
int peclr_wgrad_h(int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* slabs, int n_slabs,
                  int stride, int Ho, int Wo, const void* zeros, peclr_stream_t stream);

/* The encoder's stem (csrc/stem.hip): y = conv2d(x, W, stride 2, padding 3) with W [64][3][7][7] -- torchvision ResNet `conv1`
 * behind /root/reference/src/models/resnet_model.py:15 -- for fp32 NHWC images x [N][Hin][Win][3] -> y [N][Ho][Wo][64] NHWC,
 * Ho = (Hin - 1) / 2 + 1.  fmt 0: fp32 output at fp32 accuracy (both operands split exactly into three bf16 numbers, six
 * products, fp32 accumulation, as peclr_gemm_x6p_f32); fmt 1 / 2: bf16 / fp16 output, one product of the operands rounded
 * to that format (what autocast computes; its cast of the images rides in the kernel's staging).  `planes`: the filter in
 * fragment order, peclr_stem_pack_bytes(fmt) bytes written by peclr_stem_pack from the fp32 master weight (element strides of
 * its four dimensions given: any memory format) -- once per optimiser step.  stat_shift / stat_partial (both or neither): the
 * training statistics of y for the BatchNorm that follows, in peclr_bn2d_stats' partial layout with n_split =
 * peclr_stem_workgroups(N, Hin, Win) row blocks: stat_partial [2 n_split + 1][64].  Replaces MIOpen's 7x7 forward (and,
 * with the statistics, the last peclr_bn2d_stats pass of the step). */
int peclr_stem_pack_bytes(int fmt);
int peclr_stem_pack(const float* w, long long stride_n, long long stride_c, long long stride_h, long long stride_w, void* planes,
                    int fmt, peclr_stream_t stream);
int peclr_stem_workgroups(int N, int Hin, int Win);
int peclr_stem_conv7x7_s2(const float* x, int N, int Hin, int Win, const void* planes, int fmt, void* y,
                          const float* stat_shift, float* stat_partial, peclr_stream_t stream);
/* ... and its weight gradient: dW[n][kh][kw][c] = sum over (image, oh, ow) of dY[oh][ow][n] . x[2 oh - 3 + kh][2 ow - 3 + kw][c], with the
 * output pixel as the contraction index of the matrix cores (dY transposed into LDS by 2-byte stores, the image patch
 * gathered by eight ds_read_u16 per fragment).  dY [N][Ho][Wo][64]: fp32 (fmt 0: six products, fp32 accuracy) or bf16 / fp16
 * (fmt 1 / 2: one product, the images rounded to that format as the forward rounds them).  Persistent workgroups each write
 * one fp32 slab [64][224] -- column 32 kh + 4 kw + c; the kw = 7 and c = 3 columns carry no meaning -- slabs
 * [peclr_stem_wgrad_slabs(N, Hin, Win, fmt)][64][224], summed in a fixed order by peclr_slab_reduce_f32 (deterministic, unlike
 * MIOpen's atomically accumulated weight gradient: the last MIOpen kernel of the fp32 step).                                */
int peclr_stem_wgrad_slabs(int N, int Hin, int Win, int fmt);
int peclr_stem_wgrad(const float* x, const void* dY, int N, int Hin, int Win, int fmt, float* slabs, int n_slabs,
                     peclr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PECLR_HIP_H */
