// fp32 GEMM on the bf16 matrix cores: C[M,N] = A[M,K] . B[N,K]^T (+ addend[M,N]), fp32 in, fp32 out.
//
// gfx950 multiplies bf16 16x faster than fp32 (v_mfma_f32_32x32x16_bf16: 32 768 flop in 32 cycles; v_mfma_f32_32x32x2_f32:
// 4 096 flop in 64).  An fp32 number splits EXACTLY into three bf16 numbers,
//     h = bf16(x),  m = bf16(x - h),  l = (x - h) - m        (round to nearest; x == h + m + l; 8 + 8 + 8 significand bits)
// so a.b = sum of nine bf16 products; the three smallest (m.l, l.m, l.l: <= 2^-27 |a.b|, zero-mean, below ONE fp32 rounding)
// are dropped and the other six run on the bf16 MFMA with fp32 accumulation:
//     a.b ~= h.h + (h.m + m.h) + (h.l + l.h + m.m)
// Six MFMAs of 32 cycles replace eight of 64 per 32x32x16 block: 2.67x the fp32 MFMA rate at fp32 accuracy (measured
// against float64 in tests/test_hip_parity.py next to the v_mfma_f32 kernel).  Used for the GEMM-shaped fp32 work of
// the backbone: the bottleneck entry's fused input gradient (dX = dY W + dRes) and the 1x1 convolutions.
// Corner cases: an infinite operand gives NaN (inf - inf in the split) where an fp32 FMA chain would give +-inf -- both
// mean the run has diverged; parts below the bf16 denormal range (|x| < 2^-133) are flushed.
//
// Kernel: 128 x 128 tile, 4 waves x (2 x 2) MFMA tiles, K-tile = 32 fp32.  Both operands are K-contiguous ("NT"): a
// thread loads 16-byte words of 4 consecutive k, splits them in registers (cvt_pk / shift / sub: 5.5 VALU ops per element,
// ~40 % of the MFMA time, on the other pipe) and writes three bf16 planes per operand to LDS (row pitch 80 bytes:
// ds_read_b128 fragments of 8 k-values, bank-conflict free).  One 60 KiB LDS image, next K-tile staged in registers,
// 2 workgroups per CU.  Epilogue as in gemm_f32_nn128_kernel: wave-private 32 x 32 transposes through LDS so that
// every addend load / store is 16 bytes per lane.
#include <stdlib.h>

#include "common.hpp"

namespace peclr {
namespace {

typedef uint16_t bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int XM = 128, XN = 128, XK = 32;
constexpr int XLD = XK + 8;              // bf16 per LDS row: 80 bytes
constexpr int PLANE = 128 * XLD;         // bf16 per plane
constexpr int XEPL = 36;                 // floats per row of a wave's 32 x 32 transpose buffer

struct X6Args {
    const float* A;
    const float* B;
    const float* addend;
    float* out;
    int M, N, K, lda, ldb, ldo, ldd;
    int stream_out;
};

// one float4 (4 consecutive k) -> 8 bytes in each of the three planes (common.hpp split3_pk: exact, round to nearest)
__device__ __forceinline__ void split_store(bf16_t* planes, int offset, const float4& v, int plane = PLANE) {
    unsigned h[2], m[2], l[2];
    split3_pk(v.x, v.y, h[0], m[0], l[0]);
    split3_pk(v.z, v.w, h[1], m[1], l[1]);
    *reinterpret_cast<uint2*>(planes + offset) = make_uint2(h[0], h[1]);
    *reinterpret_cast<uint2*>(planes + plane + offset) = make_uint2(m[0], m[1]);
    *reinterpret_cast<uint2*>(planes + 2 * plane + offset) = make_uint2(l[0], l[1]);
}
__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// NB = MFMA tiles per wave along N: 2 -> 128 x 128 workgroup tile; 1 -> 128 x 64 for outputs that are 64 wide (layer1's
// 256 -> 64 convolutions: with the 128-wide tile half of every MFMA would multiply zeros and the HBM-bound shape
// would become MFMA-bound).
template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void gemm_x6_nt128_kernel(X6Args g) {
    constexpr int TNW = 64 * NB;
    constexpr int PLANE_B = TNW * XLD;   // NB = 1: 46 KiB of LDS and 144 VGPRs -> 3 workgroups per CU
    __shared__ __attribute__((aligned(16))) bf16_t lds[3 * PLANE + 3 * PLANE_B];   // A planes h, m, l | B planes h, m, l
    bf16_t* la = lds;
    bf16_t* lb = lds + 3 * PLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + TNW - 1) / TNW;
    const int j = blockIdx.x / 8;
    const int row_block = 8 * (j / nct) + (int)(blockIdx.x % 8);      // all column tiles of a row block on one XCD
    if (row_block * XM >= g.M) return;
    const int m0 = row_block * XM, n0 = (j % nct) * TNW;
    const int nk = (g.K + XK - 1) / XK;

    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // global -> registers: 128 rows x 32 fp32 per operand = 1024 x 16 bytes, 4 per thread and operand; the next K-tile
    // is in flight in registers while this one is multiplied (a second tile ahead changed nothing: not latency-bound).
    // Row of this thread's 16-byte words (+ 32 per rep): ds_write_b64 is serviced in contiguous 16-lane groups on 32
    // banks; a group writes two rows' 64-byte pieces, which must not share banks -- at an 80-byte pitch rows r and
    // r + 4 do not (20 * 4 = 16 mod 32 dwords), rows r and r + 1 do (2-way: every plane store twice as long; that was
    // a third of all LDS cycles).  So consecutive 8-thread groups take rows r, r + 4, r + 1, r + 5, ...
    float4 ra[4], rb[2 * NB];
    const int lg = tid >> 3;
    const int lr = (lg & ~7) | ((lg & 1) << 2) | ((lg >> 1) & 3), lk = (tid & 7) * 4;
    auto gload = [&](int k0) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int row = lr + 32 * rep, k = k0 + lk;
            ra[rep] = (m0 + row < g.M && k < g.K) ? *reinterpret_cast<const float4*>(g.A + (size_t)(m0 + row) * g.lda + k)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
            if (rep < 2 * NB)
                rb[rep] = (n0 + row < g.N && k < g.K) ? *reinterpret_cast<const float4*>(g.B + (size_t)(n0 + row) * g.ldb + k)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            split_store(la, (lr + 32 * rep) * XLD + lk, ra[rep]);
            if (rep < 2 * NB) split_store(lb, (lr + 32 * rep) * XLD + lk, rb[rep], PLANE_B);
        }
    };
    auto mma_tile = [&]() {
#pragma unroll
        for (int t = 0; t < XK / 16; ++t) {
            const int ko = 16 * t + 8 * kh;
            uint4 a[2][3], b[NB][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[0][p] = *reinterpret_cast<const uint4*>(la + p * PLANE + (wm * 64 + i) * XLD + ko);
                a[1][p] = *reinterpret_cast<const uint4*>(la + p * PLANE + (wm * 64 + 32 + i) * XLD + ko);
#pragma unroll
                for (int y = 0; y < NB; ++y)
                    b[y][p] = *reinterpret_cast<const uint4*>(lb + p * PLANE_B + (wn * 32 * NB + 32 * y + i) * XLD + ko);
            }
            // smallest products first; the accumulators are independent chains
#define PECLR_X6(P, Q)                                                        \
    _Pragma("unroll") for (int y = 0; y < NB; ++y) {                          \
        acc[0][y] = mma(a[0][P], b[y][Q], acc[0][y]);                         \
        acc[1][y] = mma(a[1][P], b[y][Q], acc[1][y]);                         \
    }
            PECLR_X6(2, 0) PECLR_X6(0, 2) PECLR_X6(1, 1) PECLR_X6(1, 0) PECLR_X6(0, 1) PECLR_X6(0, 0)
#undef PECLR_X6
        }
    };
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload((kt + 1) * XK);  // in flight under this tile's MFMAs
        mma_tile();
        __syncthreads();                 // every wave is done with this K-tile's image
        if (more) {
            lstore();
            __syncthreads();
        }
    }
    // epilogue (see gemm_f32_nn128_kernel): wave-private transposes, 16 bytes per lane, two tiles' addends in flight
    float* wlds = reinterpret_cast<float*>(lds) + wave * (32 * XEPL);
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const bool vec_ok = (g.N % 4 == 0) && (g.ldo % 4 == 0) && (!g.addend || g.ldd % 4 == 0);
    auto addend_tile = [&](int a, int b, float4 (&dv)[4]) {
        const int mt = m0 + wm * 64 + a * 32, nt = n0 + wn * 32 * NB + b * 32;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = mt + er + 8 * jj, n = nt + ec;
            dv[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.addend && m < g.M && n < g.N) {
                const float* src = g.addend + (size_t)m * g.ldd + n;
                if (vec_ok) {
                    const f32x4 t = g.stream_out ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src))
                                                 : *reinterpret_cast<const f32x4*>(src);
                    dv[jj] = make_float4(t[0], t[1], t[2], t[3]);
                } else {
                    dv[jj].x = src[0];
                    if (n + 1 < g.N) dv[jj].y = src[1];
                    if (n + 2 < g.N) dv[jj].z = src[2];
                    if (n + 3 < g.N) dv[jj].w = src[3];
                }
            }
        }
    };
    auto store_tile = [&](int a, int b, const f32x16& c16, const float4 (&dv)[4]) {
        const int mt = m0 + wm * 64 + a * 32, nt = n0 + wn * 32 * NB + b * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = c16[r];
        // same wave wrote and reads: LDS operations of one wave complete in order
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = mt + er + 8 * jj, n = nt + ec;
            float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
            c.x += dv[jj].x; c.y += dv[jj].y; c.z += dv[jj].z; c.w += dv[jj].w;
            if (m < g.M && n < g.N) {
                float* dst = g.out + (size_t)m * g.ldo + n;
                if (vec_ok) {
                    const f32x4 t = {c.x, c.y, c.z, c.w};
                    if (g.stream_out) __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(dst));
                    else *reinterpret_cast<f32x4*>(dst) = t;
                } else {
                    dst[0] = c.x;
                    if (n + 1 < g.N) dst[1] = c.y;
                    if (n + 2 < g.N) dst[2] = c.z;
                    if (n + 3 < g.N) dst[3] = c.w;
                }
            }
        }
    };
    float4 d0[4], d1[4];
    if constexpr (NB == 2) {
        addend_tile(0, 0, d0);
        addend_tile(0, 1, d1);
        store_tile(0, 0, acc[0][0], d0);
        addend_tile(1, 0, d0);
        store_tile(0, 1, acc[0][1], d1);
        addend_tile(1, 1, d1);
        store_tile(1, 0, acc[1][0], d0);
        store_tile(1, 1, acc[1][1], d1);
    } else {
        addend_tile(0, 0, d0);
        addend_tile(1, 0, d1);
        store_tile(0, 0, acc[0][0], d0);
        store_tile(1, 0, acc[1][0], d1);
    }
}


// ---- "TN" variant for the 1x1 weight gradients: C[M,N] = sum_k A[k,M] . B[k,N]  (dW[Cout,Cin] = dY[R,Cout]^T X[R,Cin]),
// K = R (rows of the activation) split over gridDim.y workgroups into fp32 slabs that peclr_slab_reduce_f32 adds up in
// a fixed order (deterministic, unlike the atomically accumulated split-K of the library kernels).  Both operands have
// K as their SLOW dimension, the MFMA fragments want 8 consecutive k per lane: a thread loads a 4 (k) x 4 (columns)
// block as four 16-byte words, splits the 16 values and stores, per column and plane, the four k-values as one
// ds_write_b64 into the same [column][k] planes the NT kernel uses -- the transpose happens in the choice of registers
// to pack, and the store count is the NT kernel's.  Lanes run over 8 k-groups x 8 column chunks: a 16-lane store group
// covers two column chunks x 8 k-groups = 32 distinct banks, and a load instruction reads 8 rows x 128 contiguous bytes.
template <int NB>   // 2: 128 x 128 tile; 1: 128 x 64 (64 input channels: layer1's conv3)
__global__ __launch_bounds__(256, NB == 1 ? 3 : 2) void gemm_x6_tn128_kernel(X6Args g, int kchunk) {
    constexpr int TNW = 64 * NB;
    constexpr int PLANE_B = TNW * XLD;
    __shared__ __attribute__((aligned(16))) bf16_t lds[3 * PLANE + 3 * PLANE_B];
    bf16_t* la = lds;
    bf16_t* lb = lds + 3 * PLANE;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + TNW - 1) / TNW;
    const int m0 = (int)(blockIdx.x / nct) * XM, n0 = (int)(blockIdx.x % nct) * TNW;
    const int kbeg = blockIdx.y * kchunk, kend = min(g.K, kbeg + kchunk);
    const int nk = (kend - kbeg + XK - 1) / XK;
    float* out = g.out + (size_t)blockIdx.y * g.M * g.N;     // this split's slab, ld = N

    f32x16 acc[2][NB];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 ra[4], rb[4];                                      // rows k .. k + 3 of this thread's 4 columns
    const int kg = lane & 7, chunk = 8 * wave + (lane >> 3);   // k-group (4 rows), column chunk (4 columns)
    auto gload = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + 4 * kg + q;
            ra[q] = (k < kend && m0 + 4 * chunk < g.M) ? *reinterpret_cast<const float4*>(g.A + (size_t)k * g.lda + m0 + 4 * chunk)
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[q] = (k < kend && 4 * chunk < TNW && n0 + 4 * chunk < g.N)
                        ? *reinterpret_cast<const float4*>(g.B + (size_t)k * g.ldb + n0 + 4 * chunk)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto tstore = [&](bf16_t* planes, const float4 (&v)[4], int plane) {
        const float c[4][4] = {{v[0].x, v[1].x, v[2].x, v[3].x}, {v[0].y, v[1].y, v[2].y, v[3].y},
                               {v[0].z, v[1].z, v[2].z, v[3].z}, {v[0].w, v[1].w, v[2].w, v[3].w}};   // [column][k]
#pragma unroll
        for (int jc = 0; jc < 4; ++jc)
            split_store(planes, (4 * chunk + jc) * XLD + 4 * kg, make_float4(c[jc][0], c[jc][1], c[jc][2], c[jc][3]), plane);
    };
    auto mma_tile = [&]() {
#pragma unroll
        for (int t = 0; t < XK / 16; ++t) {
            const int ko = 16 * t + 8 * kh;
            uint4 a[2][3], b[NB][3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                a[0][p] = *reinterpret_cast<const uint4*>(la + p * PLANE + (wm * 64 + i) * XLD + ko);
                a[1][p] = *reinterpret_cast<const uint4*>(la + p * PLANE + (wm * 64 + 32 + i) * XLD + ko);
#pragma unroll
                for (int y = 0; y < NB; ++y)
                    b[y][p] = *reinterpret_cast<const uint4*>(lb + p * PLANE_B + (wn * 32 * NB + 32 * y + i) * XLD + ko);
            }
#define PECLR_X6(P, Q)                                                        \
    _Pragma("unroll") for (int y = 0; y < NB; ++y) {                          \
        acc[0][y] = mma(a[0][P], b[y][Q], acc[0][y]);                         \
        acc[1][y] = mma(a[1][P], b[y][Q], acc[1][y]);                         \
    }
            PECLR_X6(2, 0) PECLR_X6(0, 2) PECLR_X6(1, 1) PECLR_X6(1, 0) PECLR_X6(0, 1) PECLR_X6(0, 0)
#undef PECLR_X6
        }
    };
    gload(kbeg);
    tstore(la, ra, PLANE);
    if (4 * chunk < TNW) tstore(lb, rb, PLANE_B);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kbeg + (kt + 1) * XK);
        mma_tile();
        __syncthreads();
        if (more) {
            tstore(la, ra, PLANE);
            if (4 * chunk < TNW) tstore(lb, rb, PLANE_B);
            __syncthreads();
        }
    }
    // epilogue: wave-private 32 x 32 transposes, 16-byte stores into the slab
    float* wlds = reinterpret_cast<float*>(lds) + wave * (32 * XEPL);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int mt = m0 + wm * 64 + a * 32, nt = n0 + wn * 32 * NB + b * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = acc[a][b][r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = mt + er + 8 * jj, n = nt + ec;
                const float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
                if (m < g.M && n < g.N) *reinterpret_cast<float4*>(out + (size_t)m * g.N + n) = c;
            }
        }
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_gemm_x6_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                 const float* addend, int ldd, peclr_stream_t stream) {
    if (!A || !B || !C) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return PECLR_ERR_SHAPE;
    if (K % 4 || lda % 4 || ldb % 4 || lda < K || ldb < K || ldc < N || (addend && ldd < N)) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(B) || !aligned16(C) || (addend && !aligned16(addend))) return PECLR_ERR_ALIGN;
    X6Args g;
    g.A = A; g.B = B; g.addend = addend; g.out = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldo = ldc; g.ldd = ldd;
    g.stream_out = (size_t)M * N * sizeof(float) > ((size_t)64 << 20);
    const int nrb = (M + XM - 1) / XM;
    if (N <= 64)       // a 64-wide output: the 128 x 64 tile (no half-empty MFMAs)
        hipLaunchKernelGGL(gemm_x6_nt128_kernel<1>, dim3(8 * ((nrb + 7) / 8)), dim3(256), 0, static_cast<hipStream_t>(stream), g);
    else
        hipLaunchKernelGGL(gemm_x6_nt128_kernel<2>, dim3(8 * ((nrb + 7) / 8) * ((N + XN - 1) / XN)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), g);
    return launch_status();
}

// Split of the K (row) range for peclr_gemm_x6_tn_f32: enough slabs that tiles x slabs fill the chip (~2 workgroups
// per CU), at least 8 K-tiles per workgroup.
extern "C" int peclr_gemm_x6_tn_slabs(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const int tnw = N <= 64 ? 64 : XN;
    const long tiles = (long)((M + XM - 1) / XM) * ((N + tnw - 1) / tnw);
    long s = (512 + tiles - 1) / tiles;
    const long max_s = (K + 8 * XK - 1) / (8 * XK);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const int kchunk = (int)(((K + s - 1) / s + XK - 1) / XK * XK);
    return (K + kchunk - 1) / kchunk;
}

extern "C" int peclr_gemm_x6_tn_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* slabs,
                                    int n_slabs, peclr_stream_t stream) {
    if (!A || !B || !slabs) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || n_slabs < 1) return PECLR_ERR_SHAPE;
    if (M % 4 || N % 4 || lda % 4 || ldb % 4 || lda < M || ldb < N) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(B) || !aligned16(slabs)) return PECLR_ERR_ALIGN;
    if (n_slabs != peclr_gemm_x6_tn_slabs(M, N, K)) return PECLR_ERR_WORKSPACE;
    X6Args g;
    g.A = A; g.B = B; g.addend = nullptr; g.out = slabs;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldo = N; g.ldd = 0; g.stream_out = 0;
    const int kchunk = ((K + n_slabs - 1) / n_slabs + XK - 1) / XK * XK;
    if (N <= 64)
        hipLaunchKernelGGL(gemm_x6_tn128_kernel<1>, dim3((M + XM - 1) / XM, n_slabs), dim3(256), 0, static_cast<hipStream_t>(stream), g, kchunk);
    else
        hipLaunchKernelGGL(gemm_x6_tn128_kernel<2>, dim3(((M + XM - 1) / XM) * ((N + XN - 1) / XN), n_slabs), dim3(256), 0,
                           static_cast<hipStream_t>(stream), g, kchunk);
    return launch_status();
}
