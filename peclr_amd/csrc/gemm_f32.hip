// K1 / K2-GEMM and the backward GEMMs of the projection head (simclr_model.py:22-33) as ONE
// templated fp32 MFMA kernel.
//
// Roofline: fp32-in MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD, 157 TF chip peak).  At the
// head's sizes (M = 2N <= a few hundred rows) the problem is a handful of 64x64 tiles, so the
// first Linear (K = Din = 2048) is split along K into slabs whose reduction is fused into the
// consumer (BN kernel) -- no atomics, deterministic order.
//
// Tiling: 256-thread workgroup = 2x2 waves, each wave one 32x32 accumulator (16 VGPRs);
// workgroup tile 64x64, BK = 32, register-staged double-buffered LDS (one barrier per K-tile).
// Operand LDS images, chosen by which dimension is contiguous in HBM so that every global
// load is a coalesced float4:
//   K-contiguous operand  -> [64 rows][BK+4]   read as ds_read_b128 (4 k-values per lane)
//   M/N-contiguous operand-> [BK rows][64+4]   read as ds_read_b32  (lanes along m)
// The k index inside a K-tile is PERMUTED consistently on both operands
// (k = 8t + 4*(lane>>5) + e for MFMA e of group t), which is free for a contraction and lets
// the K-contiguous image be read 16 bytes at a time; both read patterns are bank-conflict
// free (row stride 36 dwords under the 64-bank b128 rule, 68 dwords under the 32-bank b32
// rule).
#include <cstdlib>

#include "common.hpp"

namespace peclr {
namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int LDK = BK + 4;  // floats
constexpr int LDM = 64 + 4;  // floats
constexpr int TILE_FLOATS = 64 * LDK;  // 2304 >= 32 * LDM = 2176

struct GemmArgs {
    const float* A;
    const float* B;
    float* out;
    const float* bias;
    const float* addend;  // optional [M][N] matrix added in the epilogue (row stride ldd)
    int M, N, K, lda, ldb, ldo, ldd, kchunk;
    int stream_out;       // output (and addend) larger than the caches: non-temporal epilogue
    int flat_tiles;       // < 8 row blocks: plain tile order instead of the XCD-aware one (64 x 64 kernel)
    size_t slab_stride;  // 0 when writing C directly
};

// Global -> registers for one 64 x BK operand tile (2 float4 per thread).
template <bool KC>
__device__ __forceinline__ void tile_load(const float* __restrict__ P, int ld, int row0, int rows,
                                          int k0, int kend, int tid, float4 (&r)[2]) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KC) {
            const int row = row0 + (tid >> 3) + 32 * rep;
            const int k = k0 + (tid & 7) * 4;
            if (row < rows && k < kend) v = *reinterpret_cast<const float4*>(P + (size_t)row * ld + k);
        } else {
            const int k = k0 + (tid >> 4) + 16 * rep;
            const int row = row0 + (tid & 15) * 4;
            if (k < kend && row < rows) v = *reinterpret_cast<const float4*>(P + (size_t)k * ld + row);
        }
        r[rep] = v;
    }
}

template <bool KC>
__device__ __forceinline__ void tile_store(float* tile, int tid, const float4 (&r)[2]) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        if (KC)
            *reinterpret_cast<float4*>(tile + ((tid >> 3) + 32 * rep) * LDK + (tid & 7) * 4) = r[rep];
        else
            *reinterpret_cast<float4*>(tile + ((tid >> 4) + 16 * rep) * LDM + (tid & 15) * 4) = r[rep];
    }
}

// Fragment of k-group t (4 k-values) for the lane's row `row` (0..63 inside the tile).
template <bool KC>
__device__ __forceinline__ float4 frag(const float* tile, int row, int t, int kh) {
    if (KC) return *reinterpret_cast<const float4*>(tile + row * LDK + 8 * t + 4 * kh);
    const float* p = tile + (8 * t + 4 * kh) * LDM + row;
    return make_float4(p[0], p[LDM], p[2 * LDM], p[3 * LDM]);
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float lds[2][2][TILE_FLOATS];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, kh = lane >> 5;
    // XCD-aware tile order: the grid is 1-D (x), 8 * ceil(row blocks / 8) * column tiles.  Workgroups go to
    // the 8 XCDs round-robin, so hardware block b runs row block 8 * (j / nct) + b % 8, column tile j % nct
    // with j = b / 8: every column tile of a row block lands on the SAME XCD, one after the other, and the
    // 64 x K operand tile they share is read from HBM once and from that XCD's L2 afterwards.
    // With fewer than 8 row blocks (the projection head: M = 256 rows = 4 row blocks) that order would leave XCDs
    // idle -- there the tiles are simply dealt out one by one (g.flat_tiles), which spreads them over all XCDs.
    const int nct = (g.N + BN - 1) / BN;
    int row_block, col_tile;
    if (g.flat_tiles) {
        row_block = blockIdx.x / nct;
        col_tile = blockIdx.x % nct;
    } else {
        const int j = blockIdx.x / 8;
        row_block = 8 * (j / nct) + (int)(blockIdx.x % 8);
        col_tile = j % nct;
    }
    if (row_block * BM >= g.M) return;
    const int m0 = row_block * BM, n0 = col_tile * BN;
    const int kbeg = blockIdx.z * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    float4 ra[2], rb[2];
    if (nk > 0) {
        tile_load<A_KC>(g.A, g.lda, m0, g.M, kbeg, kend, tid, ra);
        tile_load<B_KC>(g.B, g.ldb, n0, g.N, kbeg, kend, tid, rb);
        tile_store<A_KC>(lds[0][0], tid, ra);
        tile_store<B_KC>(lds[0][1], tid, rb);
    }
    __syncthreads();
    // epilogue addend (conv1x1 dgrad + residual gradient): fetched while the LAST K-tile is being multiplied,
    // so the round trip to HBM is not exposed between the last MFMA and the first store
    float dv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) dv[r] = 0.f;
    const int n_out = n0 + wn * 32 + i;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            const int k0 = kbeg + (kt + 1) * BK;
            tile_load<A_KC>(g.A, g.lda, m0, g.M, k0, kend, tid, ra);
            tile_load<B_KC>(g.B, g.ldb, n0, g.N, k0, kend, tid, rb);
        } else if (g.addend && n_out < g.N) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + mfma32_row(r, kh);
                if (m < g.M) dv[r] = __builtin_nontemporal_load(g.addend + (size_t)m * g.ldd + n_out);
            }
        }
        const float* ta = lds[cur][0];
        const float* tb = lds[cur][1];
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            const float4 a = frag<A_KC>(ta, wm * 32 + i, t, kh);
            const float4 b = frag<B_KC>(tb, wn * 32 + i, t, kh);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (more) {
            tile_store<A_KC>(lds[cur ^ 1][0], tid, ra);
            tile_store<B_KC>(lds[cur ^ 1][1], tid, rb);
        }
        __syncthreads();
    }

    float* out = g.out + (size_t)blockIdx.z * g.slab_stride;
    const int n = n_out;
    if (n < g.N) {
        const float bv = g.bias ? g.bias[n] : 0.f;
        if (g.stream_out) {  // tall outputs (1x1-convolution gradients): write C past the caches
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + mfma32_row(r, kh);
                if (m < g.M) __builtin_nontemporal_store(acc[r] + bv + dv[r], out + (size_t)m * g.ldo + n);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + mfma32_row(r, kh);
                if (m < g.M) out[(size_t)m * g.ldo + n] = acc[r] + bv + dv[r];
            }
        }
    }
}

// ---- 128 x 128 workgroup tile, NN layout (A [M][K] K-contiguous, B [K][N] N-contiguous): the fused
// "conv1x1 input gradient + residual gradient" GEMM of the bottleneck entries (M = N*H*W rows, K = Cmid,
// N = Cin).  Each wave owns a 64 x 64 quadrant = 2 x 2 MFMA tiles in FOUR independent accumulators:
//   * consecutive MFMAs never write the same accumulator (a dependent v_mfma_f32_32x32x2_f32 chain pays for
//     every instruction the compiler slips between two of its links; the 64 x 64 kernel above is such a chain);
//   * every A / B fragment read from the LDS feeds two MFMAs, and a K-tile (BK = 32) is 64 MFMAs = 4096 cycles
//     per wave, so the one-tile-ahead register prefetch covers the global-load latency;
//   * 32 FLOP per byte fetched into the CU instead of 16.
// One LDS image (35 KiB -> 4 workgroups per CU), two barriers per K-tile, <= 128 VGPRs.
constexpr int TM = 128, TN = 128;
constexpr int LDN128 = TN + 4;
constexpr int EPL = 36;   // row stride (floats) of a wave's 32 x 32 transpose buffer in the epilogue

__global__ __launch_bounds__(256, 4) void gemm_f32_nn128_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) float la[TM * LDK];        // [128][36]
    __shared__ __attribute__((aligned(16))) float lb[BK * LDN128];     // [32][132]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + TN - 1) / TN;
    const int j = blockIdx.x / 8;
    const int row_block = 8 * (j / nct) + (int)(blockIdx.x % 8);      // all column tiles of a row block on one XCD
    if (row_block * TM >= g.M) return;
    const int m0 = row_block * TM, n0 = (j % nct) * TN;
    const int nk = (g.K + BK - 1) / BK;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // global -> registers: A 128 x 32 (4 float4 per thread: row = tid/8 + 32*rep, k = (tid%8)*4),
    //                      B 32 x 128 (4 float4 per thread: k = tid/32 + 8*rep, n = (tid%32)*4)
    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            const int row = m0 + (tid >> 3) + 32 * rep, k = k0 + (tid & 7) * 4;
            ra[rep] = (row < g.M && k < g.K) ? *reinterpret_cast<const float4*>(g.A + (size_t)row * g.lda + k)
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
            const int kb = k0 + (tid >> 5) + 8 * rep, n = n0 + (tid & 31) * 4;
            rb[rep] = (kb < g.K && n < g.N) ? *reinterpret_cast<const float4*>(g.B + (size_t)kb * g.ldb + n)
                                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            *reinterpret_cast<float4*>(la + ((tid >> 3) + 32 * rep) * LDK + (tid & 7) * 4) = ra[rep];
            *reinterpret_cast<float4*>(lb + ((tid >> 5) + 8 * rep) * LDN128 + (tid & 31) * 4) = rb[rep];
        }
    };
    gload(0);
    lstore();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload((kt + 1) * BK);
#pragma unroll
        for (int t = 0; t < BK / 8; ++t) {
            const float4 a0 = *reinterpret_cast<const float4*>(la + (wm * 64 + i) * LDK + 8 * t + 4 * kh);
            const float4 a1 = *reinterpret_cast<const float4*>(la + (wm * 64 + 32 + i) * LDK + 8 * t + 4 * kh);
            const float* pb = lb + (8 * t + 4 * kh) * LDN128 + wn * 64 + i;
            const float4 b0 = make_float4(pb[0], pb[LDN128], pb[2 * LDN128], pb[3 * LDN128]);
            const float4 b1 = make_float4(pb[32], pb[LDN128 + 32], pb[2 * LDN128 + 32], pb[3 * LDN128 + 32]);
#define PECLR_MMA4(E)                                                                   \
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b0.E, acc[0][0], 0, 0, 0);   \
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.E, b1.E, acc[0][1], 0, 0, 0);   \
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b0.E, acc[1][0], 0, 0, 0);   \
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.E, b1.E, acc[1][1], 0, 0, 0);
            PECLR_MMA4(x) PECLR_MMA4(y) PECLR_MMA4(z) PECLR_MMA4(w)
#undef PECLR_MMA4
        }
        __syncthreads();                 // every wave is done with this K-tile's image
        if (more) {
            lstore();
            __syncthreads();
        }
    }
    // Epilogue.  The MFMA accumulator layout gives a lane ONE column of a 32 x 32 tile (16 rows of it), so a
    // direct epilogue moves 4 bytes per lane and instruction -- too few bytes in flight to stream the residual
    // gradient in and the result out at HBM rate.  Each wave therefore transposes its tiles through its share of
    // the (now idle) LDS: a lane then owns 4 consecutive columns of 4 rows, and every addend load and every store
    // is 16 bytes per lane, 1 KiB per wave-instruction (8 rows x 128 bytes).
    float* wlds = la + wave * (32 * EPL);                  // 4 x 4608 bytes of `la` (18432 bytes), wave-private
    const int er = lane >> 3, ec = (lane & 7) * 4;         // this lane's row (+ 8 j) and first column in a tile
    const bool vec_ok = (g.N % 4 == 0) && (g.ldo % 4 == 0) && (!g.addend || g.ldd % 4 == 0);
    auto addend_tile = [&](int a, int b, float4 (&dv)[4]) {
        const int mt = m0 + wm * 64 + a * 32, nt = n0 + wn * 64 + b * 32;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = mt + er + 8 * jj, n = nt + ec;
            dv[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.addend && m < g.M && n < g.N) {
                const float* src = g.addend + (size_t)m * g.ldd + n;
                if (vec_ok) {
                    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
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
        const int mt = m0 + wm * 64 + a * 32, nt = n0 + wn * 64 + b * 32;
#pragma unroll
        for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * EPL + i] = c16[r];
        // same wave wrote and reads: LDS operations of one wave complete in order
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = mt + er + 8 * jj, n = nt + ec;
            float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * EPL + ec);
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
    // two tiles' residual gradients in flight at any time (the next tile's loads are issued before this tile's
    // transpose and stores)
    float4 d0[4], d1[4];
    addend_tile(0, 0, d0);
    addend_tile(0, 1, d1);
    store_tile(0, 0, acc[0][0], d0);
    addend_tile(1, 0, d0);
    store_tile(0, 1, acc[0][1], d1);
    addend_tile(1, 1, d1);
    store_tile(1, 0, acc[1][0], d0);
    store_tile(1, 1, acc[1][1], d1);
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int n_slabs,
                                                          size_t count4, int cols, const float* __restrict__ bias,
                                                          float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < count4; v += stride) {
        float4 s = reinterpret_cast<const float4*>(slabs)[v];
        for (int k = 1; k < n_slabs; ++k) {
            const float4 t = reinterpret_cast<const float4*>(slabs + (size_t)k * count4 * 4)[v];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        if (bias) {
            const int c = (int)((v * 4) % (size_t)cols);
            const float4 b = *reinterpret_cast<const float4*>(bias + c);
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        }
        reinterpret_cast<float4*>(out)[v] = s;
    }
}

// Many slabs, few elements (the split-K weight gradients: 64-256 slabs of 64 K - 1 M elements): one thread per 16-byte word
// walking all slabs leaves most of the chip idle (64 workgroups for a 128 x 512 gradient).  Here a workgroup owns 64
// consecutive words and its four waves each sum every fourth slab (eight loads in flight), then the four partial sums are
// added in a fixed order through LDS: 4x the workgroups, 4x shorter dependent chains.  Deterministic (fixed order per shape).
__global__ __launch_bounds__(256) void slab_reduce4_kernel(const float* __restrict__ slabs, int n_slabs, size_t count4,
                                                           int cols, const float* __restrict__ bias, float* __restrict__ out) {
    __shared__ float4 part[4][64];
    const int e = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const size_t v = (size_t)blockIdx.x * 64 + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < count4) {
        const float4* p = reinterpret_cast<const float4*>(slabs) + v;
        int k = sg;
        for (; k + 28 < n_slabs; k += 32) {
            float4 t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = p[(size_t)(k + 4 * u) * count4];
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += t[u].x; s.y += t[u].y; s.z += t[u].z; s.w += t[u].w; }
        }
        for (; k < n_slabs; k += 4) {
            const float4 t = p[(size_t)k * count4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
    }
    part[sg][e] = s;
    __syncthreads();
    if (sg == 0 && v < count4) {
        float4 r = part[0][e];
#pragma unroll
        for (int q = 1; q < 4; ++q) { r.x += part[q][e].x; r.y += part[q][e].y; r.z += part[q][e].z; r.w += part[q][e].w; }
        if (bias) {
            const int c = (int)((v * 4) % (size_t)cols);
            const float4 b = *reinterpret_cast<const float4*>(bias + c);
            r.x += b.x; r.y += b.y; r.z += b.z; r.w += b.w;
        }
        reinterpret_cast<float4*>(out)[v] = r;
    }
}

inline int kchunk_for(int K, int split_k) {
    const int per = (K + split_k - 1) / split_k;
    return ((per + BK - 1) / BK) * BK;
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_gemm_pick_split_k(int M, int N, int K) {
    if (M <= 0 || N <= 0 || K <= 0) return 1;
    const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    long s = 256 / tiles;               // aim for >= one workgroup per CU
    const long by_k = K / (2 * BK);     // keep >= 2 K-tiles per slab
    if (s > by_k) s = by_k;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    const int kc = kchunk_for(K, (int)s);
    return (K + kc - 1) / kc;           // effective number of non-empty slabs
}

namespace {
int gemm_launch(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* addend, int ldd, int split_k, float* slabs, peclr_stream_t stream);
}

extern "C" int peclr_gemm_f32(int layout, int M, int N, int K, const float* A, int lda, const float* B,
                              int ldb, float* C, int ldc, const float* bias, int split_k, float* slabs,
                              peclr_stream_t stream) {
    return gemm_launch(layout, M, N, K, A, lda, B, ldb, C, ldc, bias, nullptr, 0, split_k, slabs, stream);
}

extern "C" int peclr_gemm_add_f32(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                  float* C, int ldc, const float* addend, int ldd, peclr_stream_t stream) {
    if (!addend) return PECLR_ERR_NULL;
    if (ldd < N || !aligned16(addend)) return PECLR_ERR_SHAPE;
    return gemm_launch(layout, M, N, K, A, lda, B, ldb, C, ldc, nullptr, addend, ldd, 1, nullptr, stream);
}

namespace {
int gemm_launch(int layout, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* addend, int ldd, int split_k, float* slabs, peclr_stream_t stream) {
    if (!A || !B) return PECLR_ERR_NULL;
    if (split_k < 1) return PECLR_ERR_SHAPE;
    if (split_k == 1 && !C) return PECLR_ERR_NULL;
    if (split_k > 1 && !slabs) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0) return PECLR_ERR_SHAPE;
    if (layout < PECLR_GEMM_NT || layout > PECLR_GEMM_TN) return PECLR_ERR_UNSUPPORTED;
    const bool a_kc = layout != PECLR_GEMM_TN;
    const bool b_kc = layout == PECLR_GEMM_NT;
    // float4 loads run along each operand's contiguous dimension
    if ((a_kc ? K : M) % 4 || (b_kc ? K : N) % 4 || lda % 4 || ldb % 4) return PECLR_ERR_ALIGN;
    if (!aligned16(A) || !aligned16(B) || (bias && !aligned16(bias))) return PECLR_ERR_ALIGN;
    if (lda < (a_kc ? K : M) || ldb < (b_kc ? K : N) || (split_k == 1 && ldc < N)) return PECLR_ERR_SHAPE;

    GemmArgs g;
    g.A = A; g.B = B; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb;
    g.kchunk = kchunk_for(K, split_k);
    g.addend = addend;
    g.ldd = ldd;
    g.stream_out = split_k == 1 && (size_t)M * N * sizeof(float) > ((size_t)64 << 20);
    if (split_k == 1) { g.out = C; g.ldo = ldc; g.bias = bias; g.slab_stride = 0; }
    else { g.out = slabs; g.ldo = N; g.bias = nullptr; g.slab_stride = (size_t)M * N; }
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int nrb = (M + BM - 1) / BM, nct = (N + BN - 1) / BN;
    g.flat_tiles = nrb < 8;
    dim3 grid(g.flat_tiles ? nrb * nct : 8 * ((nrb + 7) / 8) * nct, 1, split_k), block(256);
    // PECLR_GEMM_TILE=64 pins the 64 x 64 kernel (A/B experiments); default: NN problems without split-K / bias
    // that fill the chip with 128 x 128 tiles (>= 2 per CU) take the 128 x 128 kernel -- the projection head's
    // own NN GEMMs (M = 256 rows: 32 such tiles) stay on the 64 x 64 kernel, which gives them 4x the workgroups
    static const int pin = [] { const char* e = getenv("PECLR_GEMM_TILE"); return e ? atoi(e) : 0; }();
    const int nrb128 = (M + TM - 1) / TM, nct128 = (N + TN - 1) / TN;
    if (pin != 64 && a_kc && !b_kc && split_k == 1 && !bias && (long)nrb128 * nct128 >= 512) {
        hipLaunchKernelGGL(gemm_f32_nn128_kernel, dim3(8 * ((nrb128 + 7) / 8) * nct128), block, 0, s, g);
        return launch_status();
    }
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, s, g);
    else if (a_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, s, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, s, g);
    return launch_status();
}
}  // namespace

extern "C" int peclr_slab_reduce_f32(const float* slabs, int n_slabs, int rows, int cols, const float* bias,
                                     float* out, peclr_stream_t stream) {
    if (!slabs || !out) return PECLR_ERR_NULL;
    if (n_slabs < 1 || rows <= 0 || cols <= 0) return PECLR_ERR_SHAPE;
    if (cols % 4 || !aligned16(slabs) || !aligned16(out) || (bias && !aligned16(bias))) return PECLR_ERR_ALIGN;
    const size_t count4 = (size_t)rows * cols / 4;
    if (n_slabs >= 8 && count4 <= ((size_t)1 << 22)) {      // the split-K weight gradients
        hipLaunchKernelGGL(slab_reduce4_kernel, dim3((unsigned)((count4 + 63) / 64)), dim3(256), 0, static_cast<hipStream_t>(stream),
                           slabs, n_slabs, count4, cols, bias, out);
        return launch_status();
    }
    int blocks = (int)((count4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), slabs,
                       n_slabs, count4, cols, bias, out);
    return launch_status();
}
