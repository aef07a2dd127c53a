// The entry-gradient GEMM of a residual block at the HBM-bound shapes, third generation ("streaming"):
//   C[M, N] = A[M, K] . W^T (+ the shortcut's gradient: dense, 1-bit-masked or compact stride-2) with the backward reduction of
//   the BatchNorm layer C arrives at -- K = 64 / 128 (the bottleneck width of layers 1 - 2), N = 4 K ... : 13 bytes of HBM traffic
//   per output element against 2 K / 6 flops; peclr_gemm_x6p_f32 ran these launches at 0.43 of the HBM roof for three rounds.
//
// What bounded gemm_x6p_kernel there is its workgroup structure, not any pipe: a tile's phases (operand loads -> products ->
// epilogue loads -> stores) are serial behind workgroup barriers and live on four workgroups per CU; tools/exp/tile_stream.hip
// shows the SAME tile-shaped accesses reaching 5.0 TB/s when nothing else is in the way (linear streams: 5.8).  Here the unit of
// work is a WAVE and there are no barriers at all:
//   * a wave owns 32 rows x all N columns.  Its A fragments are loaded straight from global memory in MFMA layout (lane = row,
//     eight consecutive k: two 16-byte loads per k-step), split once in registers (three bf16 planes) and stay there for the
//     whole row block -- no LDS, no other wave involved;
//   * for each 32-column tile the B fragments (the packed planes of peclr_x6_pack_f32, 1 KiB lane-linear pieces) come from L2 by
//     plain 16-byte loads (98 - 393 KiB per weight: resident in every XCD's L2), six products per k-step as in gemm_x6p (same
//     split, same order of products and k: bit-identical C);
//   * a tile's epilogue operands (addend rows, mask words, the BatchNorm layer's x rows) are requested one tile AHEAD (two
//     register sets), the 32 x 32 accumulator goes through a wave-private LDS transpose (16-byte rows), and the column sums of the BatchNorm
//     reduction accumulate in wave-private LDS across all the row blocks a wave walks;
//   * persistent waves: wave w of W walks row blocks w, w + W, ... (at any moment the chip streams one contiguous window of
//     rows); its sums are ONE row of the partial table ([W][2][N], n_split = W: fixed by the launch, deterministic).
// One wave per SIMD with up to 512 registers: four tiles' operands (40 KiB per wave, 160 KiB per CU) are in flight at any time;
// what hides the memory latency is each wave's own request queue -- with two waves of 256 registers per SIMD a single set of
// operands fitted, and the kernel ran at 3.0 TB/s (requests and work alternated instead of overlapping).
#include <stdlib.h>

#include "common.hpp"

namespace peclr {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SCHUNK = 12 * 1024;        // bytes of packed B per (128 columns, 16 k): gemm_x6p.hip
constexpr int XS = 36;                   // floats per row of the 32 x 32 transpose

struct X6SArgs {
    const float* A;
    const void* Bp;
    const float* addend;
    float* out;
    int M, N, K, lda, ldo, ldd;
    int add_h, add_w;                    // > 0: compact stride-2 addend (see gemm_x6p.hip)
    const unsigned* add_mask;            // optional 1-bit mask of the addend
    int stream_out;
    const float* bb_x;
    const float* bb_mean;
    const float* bb_invstd;
    const float* bb_ss;
    const unsigned* bb_mask;
    int bb_relu;
    float* bb_partial;                   // [waves][2][N]
};

__device__ __forceinline__ f32x16 mma6(const uint4& a, const uint4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// KS: k-steps of 16 (K = 16 KS); PD: k-steps of B fragments in flight; BB: the BatchNorm backward reduction is compiled in
template <int KS, int PD, bool BB>
__global__ __launch_bounds__(256, 1) void gemm_x6s_kernel(X6SArgs g) {
    constexpr int NMAX = 512;                            // columns of the per-workgroup constants / per-wave sums (N <= 512)
    __shared__ __attribute__((aligned(16))) float tr[4][32 * XS];
    __shared__ __attribute__((aligned(16))) float sums[BB ? 4 : 1][BB ? 2 * NMAX : 4];
    __shared__ __attribute__((aligned(16))) float cst[BB ? 4 * NMAX : 4];       // [mean | invstd | scale | shift][N]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, half = lane >> 5;
    const int er = lane >> 3, ec = (lane & 7) * 4;
    if constexpr (BB) {
        for (int c = tid; c < g.N; c += 256) {
            cst[c] = g.bb_mean[c]; cst[NMAX + c] = g.bb_invstd[c]; cst[2 * NMAX + c] = g.bb_ss[c]; cst[3 * NMAX + c] = g.bb_ss[g.N + c];
        }
        for (int c = lane; c < 2 * NMAX; c += 64) sums[wave][c] = 0.f;
        __syncthreads();                                  // (the only barrier: the constants are shared)
    }
    float* const wl = tr[wave];
    const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
    const int nrb = (g.M + 31) >> 5, nct = g.N >> 5;
    const unsigned char* const bp = static_cast<const unsigned char*>(g.Bp) + lane * 16;
    const int mw = g.N >> 5;                             // mask words per row

    // The epilogue operands of a tile (addend rows, mask words, the BatchNorm layer's x rows: 10 KiB per wave) are requested
    // several tiles AHEAD, into one of DEPTH register sets (below).
    struct Epi {
        f32x4 dv[4], xv[4];
        unsigned ab[4], mb[4];
        unsigned live;                                    // bit jj: row er + 8 jj of the tile exists
    };
    auto request = [&](Epi& e, int rb, int y) {
        const int m0 = rb << 5, nt = y << 5;
        e.live = 0u;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int mi = m0 + er + 8 * jj;
            const size_t m = (size_t)mi;
            const bool live = mi < g.M;
            e.live |= live ? 1u << jj : 0u;
            bool has = live && g.addend != nullptr;
            size_t arow = m;
            if (g.add_h && live) {                        // compact stride-2 addend: the row's entry in the half-resolution tensor
                const int w = mi % g.add_w, hq = mi / g.add_w, h = hq % g.add_h, img = hq / g.add_h;
                has = ((h | w) & 1) == 0;
                arow = ((size_t)img * (g.add_h >> 1) + (h >> 1)) * (g.add_w >> 1) + (w >> 1);
            }
            e.dv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
            e.ab[jj] = 0xfu;
            if (has) {
                const f32x4* src = reinterpret_cast<const f32x4*>(g.addend + arow * g.ldd + nt + ec);
                e.dv[jj] = g.stream_out ? __builtin_nontemporal_load(src) : *src;
                if (g.add_mask) e.ab[jj] = g.add_mask[m * mw + y] >> ec;
            }
            if constexpr (BB) {
                e.xv[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                e.mb[jj] = 0u;
                if (live) {
                    e.xv[jj] = *reinterpret_cast<const f32x4*>(g.bb_x + m * g.N + nt + ec);
                    if (g.bb_mask) e.mb[jj] = g.bb_mask[m * mw + y] >> ec;
                }
            }
        }
    };
    uint4 af[KS][3];                                      // the row block's A fragments: lane (row i, k-half), k = 16 t + 8 half ... + 7
    auto tile = [&](const Epi& e, int rb, int y) {
        const int m0 = rb << 5, nt = y << 5;
        // ---- products: B fragments of column tile y from the packed planes, PD k-steps in flight
        const unsigned char* bt = bp + (size_t)(y >> 2) * KS * SCHUNK + (y & 3) * 3 * 1024;
        uint4 bf[PD][3];
#pragma unroll
        for (int t = 0; t < PD; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[t][p] = *reinterpret_cast<const uint4*>(bt + (size_t)t * SCHUNK + p * 1024);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            const uint4 b0 = bf[t % PD][0], b1 = bf[t % PD][1], b2 = bf[t % PD][2];
            if (t + PD < KS) {
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[t % PD][p] = *reinterpret_cast<const uint4*>(bt + (size_t)(t + PD) * SCHUNK + p * 1024);
            }
            acc = mma6(af[t][2], b0, acc);                // the order of gemm_x6p_kernel: smallest products first
            acc = mma6(af[t][0], b2, acc);
            acc = mma6(af[t][1], b1, acc);
            acc = mma6(af[t][1], b0, acc);
            acc = mma6(af[t][0], b1, acc);
            acc = mma6(af[t][0], b0, acc);
        }
        // ---- wave-private transpose: a lane then owns four consecutive columns of four rows
#pragma unroll
        for (int r = 0; r < 16; ++r) wl[mfma32_row(r, half) * XS + i] = acc[r];
        float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
        f32x4 bmean, binv, bsc, bsh;
        if constexpr (BB) {
            bmean = *reinterpret_cast<const f32x4*>(cst + nt + ec);
            binv = *reinterpret_cast<const f32x4*>(cst + NMAX + nt + ec);
            bsc = *reinterpret_cast<const f32x4*>(cst + 2 * NMAX + nt + ec);
            bsh = *reinterpret_cast<const f32x4*>(cst + 3 * NMAX + nt + ec);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            f32x4 c = *reinterpret_cast<const f32x4*>(wl + (er + 8 * jj) * XS + ec);
#pragma unroll
            for (int q = 0; q < 4; ++q) c[q] += (e.ab[jj] >> q) & 1u ? e.dv[jj][q] : 0.f;
            if ((e.live >> jj) & 1u) {
                f32x4* dst = reinterpret_cast<f32x4*>(g.out + (size_t)(m0 + er + 8 * jj) * g.ldo + nt + ec);
                if (g.stream_out) __builtin_nontemporal_store(c, dst);
                else *dst = c;
                if constexpr (BB) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        bool on = true;
                        if (g.bb_relu) on = g.bb_mask ? (e.mb[jj] >> q) & 1u : fmaf(e.xv[jj][q], bsc[q], bsh[q]) > 0.f;
                        const float d = on ? c[q] : 0.f;
                        sb[q] += d;
                        sg[q] = fmaf(d, (e.xv[jj][q] - bmean[q]) * binv[q], sg[q]);
                    }
                }
            }
        }
        if constexpr (BB) {                               // the eight row groups of the wave meet; lane group 0 adds to the wave's sums
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { sb[q] += __shfl_xor(sb[q], o, 64); sg[q] += __shfl_xor(sg[q], o, 64); }
            }
            if (er == 0) {
                f32x4* s0 = reinterpret_cast<f32x4*>(sums[wave] + nt + ec);
                f32x4* s1 = reinterpret_cast<f32x4*>(sums[wave] + NMAX + nt + ec);
                f32x4 a = *s0, b = *s1;
#pragma unroll
                for (int q = 0; q < 4; ++q) { a[q] += sb[q]; b[q] += sg[q]; }
                *s0 = a;
                *s1 = b;
            }
        }
    };

    // DEPTH register sets: the operands of a tile are requested DEPTH - 1 tiles ahead.  The kernel runs ONE wave per SIMD
    // (up to 512 registers): what hides the memory latency is the depth of each wave's own request queue, not other waves.
    constexpr int DEPTH = 4;
    Epi e[DEPTH];
    f32x4 raw[KS][2];                                     // the NEXT row block's A rows, requested while the last tiles of this one run
    auto request_a = [&](int rb) {
        const int m0 = rb << 5;
        const int row = m0 + i < g.M ? m0 + i : g.M - 1;
        const float* ap = g.A + (size_t)row * g.lda + 8 * half;
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            raw[t][0] = *reinterpret_cast<const f32x4*>(ap + 16 * t);
            raw[t][1] = *reinterpret_cast<const f32x4*>(ap + 16 * t + 4);
        }
    };
    int rb = wid;
    if (rb < nrb) {
        request_a(rb);
#pragma unroll
        for (int u = 0; u < DEPTH - 1; ++u) request(e[u], rb, u);       // (N >= 128: at least four tiles per row block)
    }
    for (; rb < nrb; rb += nw) {
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            unsigned h[4], m[4], l[4];
            split3_pk(raw[t][0][0], raw[t][0][1], h[0], m[0], l[0]);
            split3_pk(raw[t][0][2], raw[t][0][3], h[1], m[1], l[1]);
            split3_pk(raw[t][1][0], raw[t][1][1], h[2], m[2], l[2]);
            split3_pk(raw[t][1][2], raw[t][1][3], h[3], m[3], l[3]);
            af[t][0] = make_uint4(h[0], h[1], h[2], h[3]);
            af[t][1] = make_uint4(m[0], m[1], m[2], m[3]);
            af[t][2] = make_uint4(l[0], l[1], l[2], l[3]);
        }
        const bool more = rb + nw < nrb;
        for (int y = 0; y < nct; y += DEPTH) {            // (N is a multiple of 128: a multiple of four 32-column tiles)
            if (y + DEPTH >= nct && more) request_a(rb + nw);
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int ya = y + u + DEPTH - 1;         // the tile DEPTH - 1 ahead, into the set the previous tile just released
                if (ya < nct) request(e[(u + DEPTH - 1) % DEPTH], rb, ya);
                else if (more) request(e[(u + DEPTH - 1) % DEPTH], rb + nw, ya - nct);
                tile(e[u], rb, y + u);
            }
        }
    }
    if constexpr (BB) {
        for (int c = lane; c < g.N; c += 64) {
            g.bb_partial[((size_t)wid * 2) * g.N + c] = sums[wave][c];
            g.bb_partial[((size_t)wid * 2 + 1) * g.N + c] = sums[wave][NMAX + c];
        }
    }
}

}  // namespace
}  // namespace peclr

using namespace peclr;

// Number of persistent waves (= rows of the BatchNorm partial table, n_split) a launch over M rows uses.
extern "C" int peclr_gemm_x6s_waves(int M) {
    if (M <= 0) return PECLR_ERR_SHAPE;
    const int nrb = (M + 31) / 32;
    const int blocks = nrb / 4 < 256 ? (nrb + 3) / 4 : 256;          // one workgroup of four waves (one per SIMD) on each of the 256 CUs
    return 4 * (blocks < 1 ? 1 : blocks);
}

extern "C" int peclr_gemm_x6s_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                                  const float* addend, int ldd, int add_h, int add_w, const unsigned* addend_mask,
                                  const peclr_bn_bwd_fuse* bb, peclr_stream_t stream) {
    if (!A || !Bp || !C) return PECLR_ERR_NULL;
    if (bb && (!bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial || ldc != N)) return PECLR_ERR_NULL;
    if (M <= 0 || (K != 64 && K != 128) || N <= 0 || N % 128 || N > 512) return PECLR_ERR_SHAPE;
    if (add_h && (!addend || addend_mask || add_h < 2 || add_w < 2 || add_h % 2 || add_w % 2 || M % (add_h * add_w))) return PECLR_ERR_SHAPE;
    if (addend_mask && !addend) return PECLR_ERR_NULL;
    if (lda % 4 || lda < K || ldc % 4 || ldc < N || (addend && (ldd % 4 || ldd < N))) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(Bp) || !aligned16(C) || (addend && !aligned16(addend))) return PECLR_ERR_ALIGN;
    X6SArgs g;
    g.A = A; g.Bp = Bp; g.addend = addend; g.out = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldo = ldc; g.ldd = ldd;
    g.add_h = add_h; g.add_w = add_w; g.add_mask = addend_mask;
    static const int nt_on = getenv("PECLR_X6P_STREAM_OUT") ? atoi(getenv("PECLR_X6P_STREAM_OUT")) : 1;
    g.stream_out = nt_on && (size_t)M * N * sizeof(float) > ((size_t)64 << 20);
    g.bb_x = bb ? bb->x : nullptr; g.bb_mean = bb ? bb->mean : nullptr; g.bb_invstd = bb ? bb->invstd : nullptr;
    g.bb_ss = bb ? bb->scale_shift : nullptr; g.bb_mask = bb ? bb->relu_mask : nullptr; g.bb_relu = bb ? bb->relu : 0;
    g.bb_partial = bb ? bb->partial : nullptr;
    const dim3 grid(peclr_gemm_x6s_waves(M) / 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (K == 64) {
        if (bb) hipLaunchKernelGGL((gemm_x6s_kernel<4, 2, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_x6s_kernel<4, 2, false>), grid, dim3(256), 0, s, g);
    } else {
        if (bb) hipLaunchKernelGGL((gemm_x6s_kernel<8, 1, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((gemm_x6s_kernel<8, 1, false>), grid, dim3(256), 0, s, g);
    }
    return launch_status();
}
