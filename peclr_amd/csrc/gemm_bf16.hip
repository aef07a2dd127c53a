// 16-bit-in / fp32-accumulate GEMM with an addend epilogue, for bf16 or fp16 (autocast) backbones:
//   C[M,N] (half) = A[M,K] (half) . B[N,K]^T (half) + D[M,N] (half),   half = bf16 (peclr_gemm_add_bf16) or IEEE
//   fp16 (peclr_gemm_add_f16: v_mfma_f32_32x32x16_f16, same tile, same LDS image, same epilogue)
// i.e. the bottleneck entry's "conv1 input gradient + residual branch gradient" (see peclr_gemm_add_f32)
// when activations and gradients are bf16: dX[R,Cin] = dY[R,Cmid] . Wt[Cin,Cmid]^T + dRes[R,Cin].
//
// Same skeleton as the fp32 kernel (gemm_f32.hip): 64 x 64 workgroup tile, 2 x 2 waves with one 32 x 32
// accumulator each, register-staged double-buffered LDS, XCD-aware tile order, addend fetched while the
// last K-tile is multiplied.  Differences: both operands are K-contiguous ("NT"), a K-tile is 64 bf16
// (128 B per row, the same 16-byte loads and the same 36-dword LDS row stride as the fp32 image, so the
// ds_read_b128 pattern stays bank-conflict free), and one v_mfma_f32_32x32x16_bf16 (gfx950) consumes a
// 16-byte fragment of 8 k-values per lane where the fp32 kernel issues four 32x32x2 MFMAs.
// At these shapes the kernel is HBM-bound (bf16 MFMA peak is ~16x the fp32 one): what matters is that the
// addend is read and the output written exactly once.
#include "common.hpp"

namespace peclr {
namespace {

typedef uint16_t bf16_t;   // storage type of BOTH 16-bit formats in this file
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 64, BN = 64, BKH = 64;     // BKH: bf16 elements per K-tile
constexpr int LDH = BKH + 8;                  // 72 bf16 = 36 dwords per LDS row
constexpr int TILE_H = 64 * LDH;

struct GemmHArgs {
    const bf16_t* A;
    const bf16_t* B;
    const bf16_t* addend;
    bf16_t* out;
    int M, N, K, lda, ldb, ldo, ldd;
};

// 64 rows x 64 bf16: 8 x 16-byte words per row, 512 words, 2 per thread
__device__ __forceinline__ void tile_load_h(const bf16_t* __restrict__ P, int ld, int row0, int rows, int k0, int kend,
                                            int tid, uint4 (&r)[2]) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int row = row0 + (tid >> 3) + 32 * rep;
        const int k = k0 + (tid & 7) * 8;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < rows && k < kend) v = *reinterpret_cast<const uint4*>(P + (size_t)row * ld + k);
        r[rep] = v;
    }
}
__device__ __forceinline__ void tile_store_h(bf16_t* tile, int tid, const uint4 (&r)[2]) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep)
        *reinterpret_cast<uint4*>(tile + ((tid >> 3) + 32 * rep) * LDH + (tid & 7) * 8) = r[rep];
}

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {  // round to nearest even
    const unsigned u = __float_as_uint(f);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}

// per-format pieces: MFMA, 16-bit -> fp32, fp32 -> 16-bit (round to nearest even)
struct BF16 {
    static __device__ __forceinline__ f32x16 mma(const bf16_t* pa, const bf16_t* pb, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(pa), *reinterpret_cast<const bf16x8*>(pb),
                                                      acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float up(unsigned lo16) { return __uint_as_float(lo16 << 16); }
    static __device__ __forceinline__ unsigned down(float f) { return f32_to_bf16(f); }
};
struct F16 {
    static __device__ __forceinline__ f32x16 mma(const bf16_t* pa, const bf16_t* pb, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(pa), *reinterpret_cast<const f16x8*>(pb),
                                                     acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float up(unsigned lo16) {
        const unsigned short h = (unsigned short)lo16;
        return (float)__builtin_bit_cast(_Float16, h);
    }
    static __device__ __forceinline__ unsigned down(float f) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)f); }
};

template <typename H>
__global__ __launch_bounds__(256) void gemm_bf16_nt_add_kernel(GemmHArgs g) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][TILE_H];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, kh = lane >> 5;
    // XCD-aware tile order (see gemm_f32.hip): all column tiles of a row block on one XCD
    const int nct = (g.N + BN - 1) / BN;
    const int j = blockIdx.x / 8;
    const int row_block = 8 * (j / nct) + (int)(blockIdx.x % 8);
    if (row_block * BM >= g.M) return;
    const int m0 = row_block * BM, n0 = (j % nct) * BN;
    const int nk = (g.K + BKH - 1) / BKH;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    uint4 ra[2], rb[2];
    tile_load_h(g.A, g.lda, m0, g.M, 0, g.K, tid, ra);
    tile_load_h(g.B, g.ldb, n0, g.N, 0, g.K, tid, rb);
    tile_store_h(lds[0][0], tid, ra);
    tile_store_h(lds[0][1], tid, rb);
    __syncthreads();

    // epilogue through LDS: the accumulators (lane = column) are transposed so that every thread owns 8
    // consecutive columns of a row -> 16-byte addend loads and 16-byte bf16 stores, 128 contiguous bytes per
    // output row and workgroup.  The addend words are fetched while the last K-tile is multiplied.
    constexpr int LDC = 68;  // floats per row of the staging tile (64 x 68 x 4 B = 17 KB <= operand buffers)
    const int er = tid >> 3, ec = (tid & 7) * 8;       // row (+32 per rep) and first column inside the tile
    uint4 dw[2] = {make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u)};
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            tile_load_h(g.A, g.lda, m0, g.M, (kt + 1) * BKH, g.K, tid, ra);
            tile_load_h(g.B, g.ldb, n0, g.N, (kt + 1) * BKH, g.K, tid, rb);
        } else if (g.addend) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                const int m = m0 + er + 32 * rep, n = n0 + ec;
                if (m < g.M && n < g.N) dw[rep] = *reinterpret_cast<const uint4*>(g.addend + (size_t)m * g.ldd + n);
            }
        }
        const bf16_t* ta = lds[cur][0];
        const bf16_t* tb = lds[cur][1];
#pragma unroll
        for (int t = 0; t < BKH / 16; ++t) {
            acc = H::mma(ta + (wm * 32 + i) * LDH + 16 * t + 8 * kh, tb + (wn * 32 + i) * LDH + 16 * t + 8 * kh, acc);
        }
        if (more) {
            tile_store_h(lds[cur ^ 1][0], tid, ra);
            tile_store_h(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }
    float* ct = reinterpret_cast<float*>(&lds[0][0][0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) ct[(wm * 32 + mfma32_row(r, kh)) * LDC + wn * 32 + i] = acc[r];
    __syncthreads();
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int m = m0 + er + 32 * rep, n = n0 + ec;
        if (m >= g.M || n >= g.N) continue;
        const float4 c0 = *reinterpret_cast<const float4*>(ct + (er + 32 * rep) * LDC + ec);
        const float4 c1 = *reinterpret_cast<const float4*>(ct + (er + 32 * rep) * LDC + ec + 4);
        const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const unsigned d[4] = {dw[rep].x, dw[rep].y, dw[rep].z, dw[rep].w};
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lo = c[2 * k] + H::up(d[k] & 0xFFFFu), hi = c[2 * k + 1] + H::up(d[k] >> 16);
            o[k] = H::down(lo) | (H::down(hi) << 16);
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u4*>(g.out + (size_t)m * g.ldo + n));
    }
}

}  // namespace
}  // namespace peclr

using namespace peclr;

namespace {
template <typename H>
int gemm_add_half(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc, const void* addend,
                  int ldd, peclr_stream_t stream) {
    if (!A || !B || !C) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return PECLR_ERR_SHAPE;
    if (K % 8 || N % 8 || lda % 8 || ldb % 8 || ldc % 8 || (addend && ldd % 8) || lda < K || ldb < K || ldc < N ||
        (addend && ldd < N))
        return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (addend && !aligned16(addend))) return PECLR_ERR_ALIGN;
    GemmHArgs g;
    g.A = static_cast<const bf16_t*>(A);
    g.B = static_cast<const bf16_t*>(B);
    g.addend = static_cast<const bf16_t*>(addend);
    g.out = static_cast<bf16_t*>(C);
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldo = ldc; g.ldd = ldd;
    const int nrb = (M + BM - 1) / BM, nct = (N + BN - 1) / BN;
    hipLaunchKernelGGL((gemm_bf16_nt_add_kernel<H>), dim3(8 * ((nrb + 7) / 8) * nct), dim3(256), 0,
                       static_cast<hipStream_t>(stream), g);
    return launch_status();
}
}  // namespace

extern "C" int peclr_gemm_add_bf16(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                   const void* addend, int ldd, peclr_stream_t stream) {
    return gemm_add_half<BF16>(M, N, K, A, lda, B, ldb, C, ldc, addend, ldd, stream);
}

extern "C" int peclr_gemm_add_f16(int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                                  const void* addend, int ldd, peclr_stream_t stream) {
    return gemm_add_half<F16>(M, N, K, A, lda, B, ldb, C, ldc, addend, ldd, stream);
}
