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
#include <stdlib.h>

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


// ---- 128 x 128 tile variant (round 2).  At the bottleneck shapes (K = Cmid = 64..512, N = 4 K) the 64 x 64 kernel
// above is bound by OPERAND traffic from L2, not by HBM: every workgroup re-reads 2 x 64 x K operand values for
// 64 x 64 outputs (12 544 workgroups x 64 KiB = 822 MB through L2 for 231 MB of HBM traffic at 14 x 14).  A
// 128 x 128 tile halves that, and four accumulators per wave (2 x 2 MFMA tiles) halve the LDS reads per MFMA.
// One 36 KiB LDS image with the next K-tile staged in registers (3 workgroups per CU); the last K-tile is peeled
// so that the staging registers are dead when the addend registers become live.  Epilogue: each wave transposes
// its accumulators through its own 32 x 64 strip of the (now idle) LDS so that a lane owns 16 consecutive columns
// of a row -- two 16-byte addend loads and two 16-byte stores per lane, 128 contiguous bytes per 4 lanes.
constexpr int TMH = 128, TNH = 128;
constexpr int EPH = 68;   // floats per row of a wave's 32 x 64 transpose strip (4 x 32 x 68 x 4 B = 34 816 B <= LDS image)

template <typename H>
__global__ __launch_bounds__(256, 3) void gemm_h_nt128_add_kernel(GemmHArgs g, int stream_out) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[(TMH + TNH) * LDH];   // 36 864 bytes
    bf16_t* la = lds;
    bf16_t* lb = lds + TMH * LDH;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + TNH - 1) / TNH;
    const int j = blockIdx.x / 8;
    const int row_block = 8 * (j / nct) + (int)(blockIdx.x % 8);      // all column tiles of a row block on one XCD
    if (row_block * TMH >= g.M) return;
    const int m0 = row_block * TMH, n0 = (j % nct) * TNH;
    const int nk = (g.K + BKH - 1) / BKH;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // global -> registers: 128 rows x 64 halfs per operand = 1024 x 16 bytes, 4 per thread and operand
    uint4 ra[4], rb[4];
    const int lr = tid >> 3, lk = (tid & 7) * 8;
    auto gload = [&](int k0) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int row = lr + 32 * rep, k = k0 + lk;
            ra[rep] = (m0 + row < g.M && k < g.K) ? *reinterpret_cast<const uint4*>(g.A + (size_t)(m0 + row) * g.lda + k)
                                                  : make_uint4(0u, 0u, 0u, 0u);
            rb[rep] = (n0 + row < g.N && k < g.K) ? *reinterpret_cast<const uint4*>(g.B + (size_t)(n0 + row) * g.ldb + k)
                                                  : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            *reinterpret_cast<uint4*>(la + (lr + 32 * rep) * LDH + lk) = ra[rep];
            *reinterpret_cast<uint4*>(lb + (lr + 32 * rep) * LDH + lk) = rb[rep];
        }
    };
    auto mma_tile = [&]() {
#pragma unroll
        for (int t = 0; t < BKH / 16; ++t) {
            const bf16_t* a0 = la + (wm * 64 + i) * LDH + 16 * t + 8 * kh;
            const bf16_t* b0 = lb + (wn * 64 + i) * LDH + 16 * t + 8 * kh;
            acc[0][0] = H::mma(a0, b0, acc[0][0]);
            acc[0][1] = H::mma(a0, b0 + 32 * LDH, acc[0][1]);
            acc[1][0] = H::mma(a0 + 32 * LDH, b0, acc[1][0]);
            acc[1][1] = H::mma(a0 + 32 * LDH, b0 + 32 * LDH, acc[1][1]);
        }
    };
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt + 1 < nk; ++kt) {
        gload((kt + 1) * BKH);           // in flight under this tile's MFMAs
        mma_tile();
        __syncthreads();                 // every wave is done with this K-tile's image
        lstore();
        __syncthreads();
    }
    // last K-tile: the addend travels under its MFMAs.  Lane -> row er (+16 s) of strip p, columns ec .. ec + 15.
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const int er = lane >> 2, ec = (lane & 3) * 16;
    u4 dw[2][2][2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int m = m0 + wm * 64 + p * 32 + er + 16 * s2, n = n0 + wn * 64 + ec + 8 * h;
                u4 v = {0u, 0u, 0u, 0u};
                if (g.addend && m < g.M && n < g.N) {
                    const u4* src = reinterpret_cast<const u4*>(g.addend + (size_t)m * g.ldd + n);
                    v = stream_out ? __builtin_nontemporal_load(src) : *src;
                }
                dw[p][s2][h] = v;
            }
    mma_tile();
    __syncthreads();                     // the operand image is dead: its memory becomes the transpose strips
    float* wl = reinterpret_cast<float*>(lds) + wave * (32 * EPH);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) wl[mfma32_row(r, kh) * EPH + y * 32 + i] = acc[p][y][r];
        // same wave wrote and reads (and overwrites in the next strip): LDS operations of one wave complete in order
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int row = er + 16 * s2;
            const int m = m0 + wm * 64 + p * 32 + row;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int n = n0 + wn * 64 + ec + 8 * h;
                const float4 c0 = *reinterpret_cast<const float4*>(wl + row * EPH + ec + 8 * h);
                const float4 c1 = *reinterpret_cast<const float4*>(wl + row * EPH + ec + 8 * h + 4);
                const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                const u4 d = dw[p][s2][h];
                u4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = c[2 * k] + H::up(d[k] & 0xFFFFu), hi = c[2 * k + 1] + H::up(d[k] >> 16);
                    o[k] = H::down(lo) | (H::down(hi) << 16);
                }
                if (m < g.M && n < g.N) {
                    u4* dst = reinterpret_cast<u4*>(g.out + (size_t)m * g.ldo + n);
                    if (stream_out) __builtin_nontemporal_store(o, dst);
                    else *dst = o;
                }
            }
        }
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
    // shapes that fill the chip with 128 x 128 tiles (>= 2 per CU) take the 128 x 128 kernel (PECLR_GEMM_TILE=64 pins
    // the 64 x 64 one, as for the fp32 GEMM)
    static const int pin = [] { const char* e = getenv("PECLR_GEMM_TILE"); return e ? atoi(e) : 0; }();
    const int nrb128 = (M + TMH - 1) / TMH, nct128 = (N + TNH - 1) / TNH;
    if (pin != 64 && (long)nrb128 * nct128 >= 512) {
        const int stream_out = (size_t)M * N * 2 > ((size_t)64 << 20);   // output (and addend) larger than the caches
        hipLaunchKernelGGL((gemm_h_nt128_add_kernel<H>), dim3(8 * ((nrb128 + 7) / 8) * nct128), dim3(256), 0,
                           static_cast<hipStream_t>(stream), g, stream_out);
        return launch_status();
    }
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
