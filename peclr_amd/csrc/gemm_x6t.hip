// Weight gradients on the bf16 matrix cores at fp32 accuracy, second generation: C[M, N] = sum_k A[k, M] . B[k (+ shift), N]
// with the contraction over the ROWS of two NHWC activations (dW = dY^T X for 1x1 convolutions; for a 3x3 convolution one
// such product per filter tap, X read at the pixel the tap points at).
//
// Both operands are activations, so both have to be split (x == h + m + l, common.hpp) inside the kernel; what this
// kernel changes against gemm_x6_tn128_kernel is how often and where:
//   * 256 x 256 output tiles (8 waves as 4 x 2, each 64 x 128 = 2 x 4 MFMA tiles, six products each): an operand row is split
//     once per 256 output columns of the other operand instead of once per 128 -- 1.8 VALU instructions per MFMA, the
//     ratio of the forward kernel (gemm_x6p.hip), where the 128 x 128 kernel has 3.7.  128-wide problems take 128 x 256,
//     256 x 128 or 128 x 128 tiles (wave tiles 32 x 128, 64 x 64, 32 x 64) of the same code.
//   * k-step 16 with DOUBLE-buffered planes and one barrier per step: the split / plane stores of step t + 1 run under
//     the MFMAs of step t (the old kernel alternated a multiply phase and a store phase).
//   * a thread owns a 4 (k) x 4 (columns) block: four 16-byte loads, columns paired over k by the packed conversion
//     (the transpose costs nothing), 8-byte stores.  Plane layout [k-half][32-column tile][slot][8 k] with
//     slot(i) = (i & 3) * 8 + ((i >> 2) ^ ((i >> 1) & 1) * 4): the sixteen lanes of a store instruction (eight column chunks
//     x two k-quads) fill 128 contiguous bytes, the sixteen lanes of a fragment read hit sixteen distinct 16-byte
//     slots -- both conflict-free.
// K is split over gridDim.y workgroup rows writing fp32 slabs [split][M][ldc] that peclr_slab_reduce_f32 adds in a fixed
// order (deterministic).  taps = 9: gridDim.z = 9, B rows shifted by the tap's pixel offset (zeros outside the image),
// output columns tap * N + n (the [Cout][3][3][Cin] layout of a channels_last weight).
#include "common.hpp"

#include <type_traits>

namespace peclr {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TK = 16;

struct X6TArgs {
    const float* A;                  // [K][lda]: M columns used
    const float* B;                  // [K][ldb]: N columns used
    float* slabs;                    // [splits][M][ldc]
    int M, N, K, lda, ldb, ldc;
    int kchunk;                      // rows per split (multiple of 16)
    int taps, H, W;                  // taps = 9: the nine 3x3 filter taps; H x W: image extents of A's rows (output pixels)
    int stride;                      // 2: B's rows are the pixels of 2H x 2W images, read at (2 oh + dh, 2 ow + dw)
    const float* zeros;
};

__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ int slot_of(int i) { return (i & 3) * 8 + ((i >> 2) ^ (((i >> 1) & 1) << 2)); }

// MT x NT: 32 x 32 MFMA tiles per wave; waves WGM (M) x 8 / WGM (N): workgroup tile 32 WGM MT x 256 / WGM NT (4 x 2 waves:
// 128 MT x 64 NT; 2 x 4 waves for the 64-row gradients of layer1: 64 MT x 128 NT).  GEO: B's rows are the pixels of the
// stride-2 convolution's input (a 1x1 / stride-2 shortcut), found from the output pixel each row of A is.
// PF: k-steps of global loads in flight per thread (register stages).  (Measured on the 64-wide gradients, whose k-steps
// are short: PF = 2 changes nothing -- hipcc's wait-count pass still drains every load before the split -- and PF = 4
// costs the second workgroup per CU its registers, 262 -> 298 us.  PF = 1 everywhere.)
template <int MT, int NT, int WGM, bool GEO, int PF>
__global__ __launch_bounds__(512, 2) void gemm_x6t_kernel(X6TArgs g) {
    constexpr int WGN = 8 / WGM;
    constexpr int TM = 32 * WGM * MT, TN = 32 * WGN * NT;
    static_assert((TM + TN) / 64 <= 8 && TM % 64 == 0 && TN % 64 == 0, "one 64-column group of the split per wave");
    constexpr int HALF_A = TM * 16, HALF_B = TN * 16;         // bytes of one k-half of a plane
    constexpr int PL_A = 2 * HALF_A, PL_B = 2 * HALF_B;
    constexpr int BUF = 3 * (PL_A + PL_B);                    // one k-step of both operands
    constexpr int XEPL = 36;
    static_assert(2 * BUF >= 8 * 32 * XEPL * 4, "epilogue transposes live in the plane buffers");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + TN - 1) / TN;
    const int m0 = (int)(blockIdx.x / nct) * TM, n0 = (int)(blockIdx.x % nct) * TN;
    const int kbeg = blockIdx.y * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + TK - 1) / TK;

    // ---- this thread's block of the split: operand (A for the first TM threads' worth of blocks, then B), column chunk
    // (4 columns), k-quad (4 rows).  Lane bits: [2:0] chunk within a 32-column tile, [3] k-quad & 1, [4] k-half, [5] tile
    // parity; a wave covers 64 columns.
    const int blk_col64 = wave;                               // 64-column group index over A's then B's columns
    const bool is_a = blk_col64 < TM / 64;
    const bool active = blk_col64 < (TM + TN) / 64;
    const int cgrp = is_a ? blk_col64 : blk_col64 - TM / 64;
    const int chunk = (lane & 7) + 8 * (lane >> 5);           // 0..15 within the 64-column group
    const int kq = (lane >> 3) & 3;
    const int col = cgrp * 64 + 4 * chunk;                    // first of this thread's 4 columns inside the tile
    const float* src = is_a ? g.A + m0 + col : g.B + n0 + col;
    const int ld = is_a ? g.lda : g.ldb;
    const bool col_ok = active && (is_a ? m0 + col < g.M : n0 + col < g.N);
    // plane store address of column col + j: tile (col >> 5), slot of ((col & 31) + j), k-quad half
    const int tile = col >> 5, cin = col & 31;
    const int st_base = (kq >> 1) * (is_a ? HALF_A : HALF_B) + tile * 512 + (kq & 1) * 8 + (is_a ? 0 : 3 * PL_A);
    const int st_plane = is_a ? PL_A : PL_B;

    // rows that exactly ONE workgroup of a split reads (the operand whose other side fits one tile: both operands of layer1's
    // 256 x 64 / 64 x 256 gradients) are loaded with the non-temporal hint: a linear read runs at 6.8 TB/s with it, 4.3 without
    // (tools/exp/rw_mix.hip); `conv1x1_wgrad~hbm` 192 -> 180 us.  Rows several workgroups share keep the plain load (L2 reuse).
    const bool once = is_a ? nct == 1 : (int)gridDim.x == nct;
    f32x4 ld4[PF][4];
    // GEO: pixel (oh, ow) of the first row of this thread's next block, advanced by 16 rows per k-step (no divisions in
    // the loop); gload() is called with t = 0, 1, 2, ... in order
    int ow_t = 0, oh_t = 0;
    if constexpr (GEO) {
        const int k = kbeg + 4 * kq;
        ow_t = k % g.W;
        oh_t = (k / g.W) % g.H;
    }
    auto gload = [&](int t, f32x4 (&ld4)[4]) {                // rows kbeg + 16 t + 4 kq + q of this thread's 4 columns
        int ow = ow_t, oh = oh_t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kbeg + t * TK + 4 * kq + q;
            const float* p = src + (size_t)k * ld;
            const bool ok = col_ok && k < kend;
            if constexpr (GEO) {
                // B row = input pixel (2 oh, 2 ow) of the same image: 4 (k - oh W - ow) rows of whole images before it
                if (!is_a) p = src + (size_t)(4 * (k - oh * g.W - ow) + 4 * oh * g.W + 2 * ow) * ld;
                if (++ow == g.W) { ow = 0; oh = oh + 1 == g.H ? 0 : oh + 1; }
            }
            p = ok ? p : g.zeros;
            if (once) ld4[q] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
            else ld4[q] = *reinterpret_cast<const f32x4*>(p);
        }
        if constexpr (GEO) {
            ow_t += TK;
#pragma unroll
            for (int it = 0; it < 3; ++it)                    // W >= 7: at most three image rows per 16 pixels
                if (ow_t >= g.W) { ow_t -= g.W; oh_t = oh_t + 1 == g.H ? 0 : oh_t + 1; }
        }
    };
    auto split_store = [&](int buf, const f32x4 (&ld4)[4]) {
        if (!active) return;
        unsigned char* base = lds + buf * BUF + st_base;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h[2], m[2], l[2];
            float v[4] = {ld4[0][j], ld4[1][j], ld4[2][j], ld4[3][j]};
            split3_pk(v[0], v[1], h[0], m[0], l[0]);
            split3_pk(v[2], v[3], h[1], m[1], l[1]);
            unsigned char* d = base + slot_of(cin + j) * 16;
            *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2*>(d + st_plane) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2*>(d + 2 * st_plane) = make_uint2(l[0], l[1]);
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][y][r] = 0.f;

    const int fa = kh * HALF_A + (wm * MT) * 512 + slot_of(i) * 16;                  // + a * 512 + plane * PL_A
    const int fb = 3 * PL_A + kh * HALF_B + (wn * NT) * 512 + slot_of(i) * 16;       // + y * 512 + plane * PL_B

    // step u lives in register stage u % PF: loaded PF steps ahead, split into the planes one step ahead
    if (nk > 0) {
        gload(0, ld4[0]);
        split_store(0, ld4[0]);
#pragma unroll
        for (int u = 1; u <= PF; ++u)
            if (u < nk) gload(u, ld4[u % PF]);
    }
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += PF)
#pragma unroll
    for (int st = 0; st < PF; ++st) {
        const int t = t0 + st;
        if (t >= nk) break;                                   // (uniform)
        const unsigned char* bufp = lds + (t & 1) * BUF;
        uint4 af[MT][3];
#pragma unroll
        for (int a = 0; a < MT; ++a)
#pragma unroll
            for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const uint4*>(bufp + fa + a * 512 + p * PL_A);
        constexpr int NH = NT > 2 ? 2 : 1, YH = NT / NH;          // B fragments in two halves (registers)
#pragma unroll
        for (int half = 0; half < NH; ++half) {
            uint4 bf[YH][3];
#pragma unroll
            for (int y = 0; y < YH; ++y)
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[y][p] = *reinterpret_cast<const uint4*>(bufp + fb + (half * YH + y) * 512 + p * PL_B);
#define PECLR_X6(P, Q)                                                                        \
    _Pragma("unroll") for (int y = 0; y < YH; ++y) _Pragma("unroll") for (int a = 0; a < MT; ++a) \
        acc[a][half * YH + y] = mma(af[a][P], bf[y][Q], acc[a][half * YH + y]);
            PECLR_X6(2, 0) PECLR_X6(0, 2) PECLR_X6(1, 1) PECLR_X6(1, 0) PECLR_X6(0, 1) PECLR_X6(0, 0)
#undef PECLR_X6
            if (half == 0 && t + 1 < nk) {
                split_store((t + 1) & 1, ld4[(st + 1) % PF]);       // rows of step t + 1 -> the other buffer
                if (t + 1 + PF < nk) gload(t + 1 + PF, ld4[(st + 1) % PF]);
            }
        }
        __syncthreads();
    }

    // epilogue: wave-private 32 x 32 transposes through LDS, 16-byte stores into this split's slab
    float* out = g.slabs + (size_t)blockIdx.y * g.M * g.ldc;
    float* wlds = reinterpret_cast<float*>(lds) + wave * (32 * XEPL);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int mt = m0 + (wm * MT + a) * 32, nt = n0 + (wn * NT + y) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = acc[a][y][r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = mt + er + 8 * jj, n = nt + ec;
                const float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
                if (m < g.M && n < g.N) *reinterpret_cast<float4*>(out + (size_t)m * g.ldc + n) = c;
            }
        }
}

// ---- 3x3 weight gradient, all nine taps in one workgroup.  With one product per tap (gemm_x6t_kernel<.., 9>) the dY
// operand is split nine times over; here a workgroup (8 waves as 4 x 2) owns a 128 (Cout) x 64 (Cin) block of ALL nine taps:
// per k-step dY's 128 columns are split ONCE, X's 64 columns once per tap (rows shifted by the tap, zeros outside the
// image), and every wave runs 9 x 6 MFMAs into nine 32 x 32 accumulators (144 registers) -- 2.2 VALU instructions per MFMA
// instead of 3.7.  Eleven 64-column groups of planes per k-step (dY: 2, X: one per tap), 6 KiB each, double-buffered
// (132 KiB); waves 0-2 split two groups per step, the others one.
// WGM = 2 (the 64-channel convolutions of layer1): a 64 (Cout) x 64 (Cin) block, one group of dY, waves 2 x 2 x two tap
// halves (taps 0-4 and 5-8: five and four accumulators).
// stride 2 (the first block of layers 2-4): X's rows are the pixels of the 2H x 2W input, read at (2 oh + dh, 2 ow + dw).
// ABL: ablation switches for tools/exp/x6w_ablate.hip (0 in the library): 1 = X groups stored without the split (raw halves as
// "planes"), 2 = dY groups likewise, 4 = no plane stores after the first k-step, 8 = no global loads after the prologue
template <int WGM, int ABL = 0>
__global__ __launch_bounds__(512, 2) void gemm_x6w_kernel(X6TArgs g) {
    constexpr int GRP = 3 * 2 * 64 * 16;                      // bytes of one 64-column group: [plane][k-half][tile][slot][16]
    constexpr int NGA = WGM / 2, NG = NGA + 9, BUF = NG * GRP;
    constexpr int NACC = WGM == 4 ? 9 : 5;
    constexpr int XEPL = 36;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WGM == 4 ? wave >> 1 : (wave >> 1) & 1, wn = wave & 1;
    const int tap0 = WGM == 4 ? 0 : (wave >> 2) * 5;          // first tap of this wave
    const int ntap = WGM == 4 ? 9 : (wave >> 2 ? 4 : 5);
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + 63) / 64;
    const int m0 = (int)(blockIdx.x / nct) * (32 * WGM), n0 = (int)(blockIdx.x % nct) * 64;
    const int sh2 = g.stride == 2;
    const int kbeg = blockIdx.y * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + TK - 1) / TK;

    const int chunk = (lane & 7) + 8 * (lane >> 5), kq = (lane >> 3) & 3;
    const int cin = 4 * (chunk & 7), tile = chunk >> 3;
    const int st_off = (kq >> 1) * 1024 + tile * 512 + (kq & 1) * 8;     // inside a group's plane
    // group gi: 0 .. NGA - 1 = dY columns m0 + 64 gi ...; NGA + tap = X columns n0 ..., rows shifted by the tap
    const int g1 = wave, g2 = wave + 8;                        // this wave's groups (g2 only for the first NG - 8 waves)
    const bool two = g2 < NG;
    struct Src { const float* p; int ld, dh, dw; bool ok, is_a; };
    auto src_of = [&](int gi) {
        Src s;
        s.is_a = gi < NGA;
        const int tap = gi - NGA;
        s.dh = s.is_a ? 0 : tap / 3 - 1;
        s.dw = s.is_a ? 0 : tap % 3 - 1;
        const int c = 4 * chunk + (s.is_a ? 64 * gi : 0);
        s.ld = s.is_a ? g.lda : g.ldb;
        s.p = s.is_a ? g.A + m0 + c : g.B + n0 + c;
        s.ok = s.is_a ? m0 + c < g.M : n0 + c < g.N;
        return s;
    };
    const Src s1 = src_of(g1), s2 = src_of(two ? g2 : g1);
    int ow_t, oh_t;
    {
        const int k = kbeg + 4 * kq;
        ow_t = k % g.W;
        oh_t = (k / g.W) % g.H;
    }
    f32x4 la[4], lb[4];
    auto gload1 = [&](const Src& s, int t, f32x4 (&r)[4]) {
        int ow = ow_t, oh = oh_t;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kbeg + t * TK + 4 * kq + q;
            bool ok = s.ok && k < kend;
            const int ih = (oh << sh2) + s.dh, iw = (ow << sh2) + s.dw;          // the input pixel the tap points at
            if (!s.is_a) ok = ok && (unsigned)ih < (unsigned)(g.H << sh2) && (unsigned)iw < (unsigned)(g.W << sh2);
            // stride 2: 4 (k - oh W - ow) rows of whole input images before this one's (ih, iw)
            const long row = s.is_a ? (long)k : sh2 ? 4L * (k - oh * g.W - ow) + ih * 2 * g.W + iw : (long)k + s.dh * g.W + s.dw;
            const float* p = s.p + row * s.ld;
            r[q] = *reinterpret_cast<const f32x4*>(ok ? p : g.zeros);
            if (++ow == g.W) { ow = 0; oh = oh + 1 == g.H ? 0 : oh + 1; }
        }
    };
    auto gload = [&](int t) {
        gload1(s1, t, la);
        if (two) gload1(s2, t, lb);
        ow_t += TK;
#pragma unroll
        for (int it = 0; it < 3; ++it)
            if (ow_t >= g.W) { ow_t -= g.W; oh_t = oh_t + 1 == g.H ? 0 : oh_t + 1; }
    };
    auto store1 = [&](int buf, int gi, const f32x4 (&r)[4]) {
        unsigned char* base = lds + buf * BUF + gi * GRP + st_off;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            unsigned h[2], m[2], l[2];
            if ((ABL & 1) && gi >= NGA || (ABL & 2) && gi < NGA) {
                h[0] = __float_as_uint(r[0][j]); m[0] = __float_as_uint(r[1][j]); l[0] = h[0] ^ m[0];
                h[1] = __float_as_uint(r[2][j]); m[1] = __float_as_uint(r[3][j]); l[1] = h[1] ^ m[1];
            } else {
                split3_pk(r[0][j], r[1][j], h[0], m[0], l[0]);
                split3_pk(r[2][j], r[3][j], h[1], m[1], l[1]);
            }
            unsigned char* d = base + slot_of(cin + j) * 16;
            *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
            *reinterpret_cast<uint2*>(d + 2048) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2*>(d + 4096) = make_uint2(l[0], l[1]);
        }
    };
    auto split_store = [&](int buf) {
        store1(buf, g1, la);
        if (two) store1(buf, g2, lb);
    };

    f32x16 acc[NACC];
#pragma unroll
    for (int tp = 0; tp < NACC; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
    const int fa = (wm >> 1) * GRP + kh * 1024 + (wm & 1) * 512 + slot_of(i) * 16;       // + plane * 2048
    const int fb = (NGA + tap0) * GRP + kh * 1024 + wn * 512 + slot_of(i) * 16;           // + tap * GRP + plane * 2048

    if (nk > 0) {
        gload(0);
        split_store(0);
        if (nk > 1) gload(1);
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const unsigned char* bufp = lds + (t & 1) * BUF;
        uint4 af[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const uint4*>(bufp + fa + p * 2048);
#pragma unroll
        for (int tp = 0; tp < NACC; ++tp) {
            if (WGM == 4 || tp < ntap) {                      // (wave-uniform: the second tap half has four taps)
                uint4 bf[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[p] = *reinterpret_cast<const uint4*>(bufp + fb + tp * GRP + p * 2048);
                acc[tp] = mma(af[2], bf[0], acc[tp]);
                acc[tp] = mma(af[0], bf[2], acc[tp]);
                acc[tp] = mma(af[1], bf[1], acc[tp]);
                acc[tp] = mma(af[1], bf[0], acc[tp]);
                acc[tp] = mma(af[0], bf[1], acc[tp]);
                acc[tp] = mma(af[0], bf[0], acc[tp]);
            }
            if (tp == (WGM == 4 ? 2 : 1) && t + 1 < nk) {
                if constexpr (!(ABL & 4)) split_store((t + 1) & 1);
                if constexpr (!(ABL & 8)) if (t + 2 < nk) gload(t + 2);
            }
        }
        __syncthreads();
    }

    float* out = g.slabs + (size_t)blockIdx.y * g.M * g.ldc;
    float* wlds = reinterpret_cast<float*>(lds) + wave * (32 * XEPL);
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const int mt = m0 + wm * 32, nt = n0 + wn * 32;
#pragma unroll
    for (int tp = 0; tp < NACC; ++tp) {
        if (WGM != 4 && tp >= ntap) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = acc[tp][r];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = mt + er + 8 * jj, n = nt + ec;
            const float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
            if (m < g.M && n < g.N) *reinterpret_cast<float4*>(out + (size_t)m * g.ldc + (tap0 + tp) * g.N + n) = c;
        }
    }
}

// ---- the same product (3x3 / stride 1, all nine taps in one workgroup, same LDS layout, same order of MFMAs per accumulator:
// bit-identical slabs) with the loop re-scheduled (round 6).  In gemm_x6w_kernel hipcc leaves a k-step as "54 MFMAs, then ~550
// vector instructions" -- address arithmetic (64-bit row pointers, per-row bounds tests through exec-mask branches), splits,
// plane stores -- in blocks that hold no MFMA, and the workgroup's barrier keeps the two waves of a SIMD in phase, so the matrix
// cores idle through them (tools/exp/x6w_ablate.hip: the loop without that work runs at 0.70 of the six-product peak, with it
// at 0.45).  Here
//   * rows are fetched by BUFFER loads: 32-bit offsets (one add per row), a row that does not exist -- before the tensor, past
//     its end, or masked -- is an out-of-range offset and reads as zeros: no pointer selects, no branches;
//   * whether the tap's neighbour of an output pixel lies inside the image comes from a table in LDS (per tap and pixel of an
//     image: 4 bits for the pixel and its three successors), built once per workgroup, instead of (oh, ow) arithmetic per row;
//   * the step is ONE basic block in which the split / store / load-issue work of the NEXT steps is cut into half-units (two
//     rows of one 64-column group: 4 splits, 12 4-byte plane stores, 2 loads) placed between the MFMAs of taps 0 - 7
//     (sched_barrier fences keep hipcc from clustering them again); fragments of tap t + 1 are read under the MFMAs of tap t.
// Waves 0 .. NG - 9 own two column groups, the others one: the loop is instantiated for both (TWO) and chosen per wave.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int X6W2_MAX_HW = 3136;                             // table bytes per tap (an image of up to 56 x 56 output pixels)
constexpr int X6W2_TAB = 9 * X6W2_MAX_HW + 192;               // bytes of the table area
constexpr int X6W2_S2_MAX_HW = X6W2_TAB / 36 - 3;             // stride 2: four bytes per tap and pixel (+ 3 wrap entries per tap): 28 x 28

// S2: the 3x3 / STRIDE-2 weight gradient (dY over the H x W output pixels, X over the 2H x 2W input pixels, read at (2 oh + dh,
// 2 ow + dw)).  The row of X a (tap, output pixel) reads is not linear in the output pixel, so the table holds, per tap and pixel
// of an image, the BYTE OFFSET of that row inside the image (or 2^30: no such row) -- with three more entries per tap for the quads
// that run into the next image -- and a lane's offset is (its image's base + its columns) + the entry.

template <int WGM, bool S2 = false>
__global__ __launch_bounds__(512, 2) void gemm_x6w2_kernel(X6TArgs g) {
    constexpr int GRP = 3 * 2 * 64 * 16;                      // bytes of one 64-column group: [plane][k-half][tile][slot][16]
    constexpr int NGA = WGM / 2, NG = NGA + 9, BUF = NG * GRP;
    constexpr int NACC = WGM == 4 ? 9 : 5;
    constexpr int XEPL = 36;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF + X6W2_TAB];
    unsigned char* const tab = lds + 2 * BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WGM == 4 ? wave >> 1 : (wave >> 1) & 1, wn = wave & 1;
    const int tap0 = WGM == 4 ? 0 : (wave >> 2) * 5;          // first tap of this wave
    const int ntap = WGM == 4 ? 9 : (wave >> 2 ? 4 : 5);
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + 63) / 64;
    const int m0 = (int)(blockIdx.x / nct) * (32 * WGM), n0 = (int)(blockIdx.x % nct) * 64;
    const int kbeg = blockIdx.y * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + TK - 1) / TK;
    const int HW = g.H * g.W;

    const unsigned s2_ld4 = (unsigned)g.ldb * 4u, s2_img = 4u * HW * s2_ld4;      // S2: bytes per row / per image of X
    const int s2_row = (HW + 3) * 4;                                               // S2: bytes of a tap's table row
    if constexpr (S2) {
        unsigned* t32 = reinterpret_cast<unsigned*>(tab);
        for (int p = tid; p < HW; p += 512) {
            const int oh = p / g.W, ow = p - oh * g.W;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ih = 2 * oh + tap / 3 - 1, iw = 2 * ow + tap % 3 - 1;          // (never past the bottom / right edge)
                const bool in = ih >= 0 && iw >= 0;
                const unsigned e = in ? (unsigned)(ih * 2 * g.W + iw) * s2_ld4 : 0x40000000u;
                t32[tap * (HW + 3) + p] = e;
                if (p < 3) t32[tap * (HW + 3) + HW + p] = in ? e + s2_img : 0x40000000u;
            }
        }
        __syncthreads();
    } else {
    // the table: tab[tap][p] bit q = the tap's neighbour of pixel (p + q) mod HW of an image lies inside it.  First the nine
    // taps of every pixel (one division per pixel; 16 bits each, in the plane buffers, which nothing uses yet), then the quads

        unsigned short* pm = reinterpret_cast<unsigned short*>(lds);
        for (int p = tid; p < HW; p += 512) {
            const int oh = p / g.W, ow = p - oh * g.W;
            const unsigned rv = (oh > 0 ? 1u : 0u) | 2u | (oh + 1 < g.H ? 4u : 0u), cv = (ow > 0 ? 1u : 0u) | 2u | (ow + 1 < g.W ? 4u : 0u);
            unsigned bits = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) bits |= ((rv >> (tap / 3)) & (cv >> (tap % 3)) & 1u) << tap;
            pm[p] = (unsigned short)bits;
        }
        __syncthreads();
        for (int p = tid; p < HW; p += 512) {
            unsigned v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = pm[p + q < HW ? p + q : p + q - HW];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                tab[tap * X6W2_MAX_HW + p] = (unsigned char)(((v[0] >> tap) & 1u) | ((v[1] >> tap) & 1u) << 1 | ((v[2] >> tap) & 1u) << 2 | ((v[3] >> tap) & 1u) << 3);
        }
        __syncthreads();
    }

    const int chunk = (lane & 7) + 8 * (lane >> 5), kq = (lane >> 3) & 3;
    const int cin = 4 * (chunk & 7), tile = chunk >> 3;
    const int st_off = (kq >> 1) * 1024 + tile * 512 + (kq & 1) * 8;     // inside a group's plane
    // group gi: 0 .. NGA - 1 = dY columns m0 + 64 gi ...; NGA + tap = X columns n0 ..., rows shifted by the tap
    struct Grp {
        __amdgpu_buffer_rsrc_t rs;       // (wave-uniform) the operand, cut off after the last row this group may read
        unsigned off;                    // byte offset of row k0 + shift, this lane's four columns (columns that do not exist: 2^31 + ...,
                                         // out of range whatever is added)
        unsigned step, ld4;              // (wave-uniform) bytes per k-step / per row
        int tb;                          // (wave-uniform) LDS offset of this group's table row (dY: the centre tap's, all ones)
        int st;                          // LDS offset of this lane's stores inside buffer 0
        bool isx;                        // (wave-uniform) S2: a group of X (offsets through the table)
    };
    auto grp_of = [&](int gi) {
        Grp s;
        const bool is_a = gi < NGA;
        const int tap = is_a ? 4 : gi - NGA;
        const int shift = (tap / 3 - 1) * g.W + (tap % 3 - 1);
        const int ld = is_a ? g.lda : g.ldb;
        const int c = 4 * chunk + (is_a ? 64 * gi : 0);
        const bool ok = is_a ? m0 + c < g.M : n0 + c < g.N;
        const long rows = S2 ? (is_a ? (long)g.K : 4L * g.K) : (long)g.K + (shift < 0 ? shift : 0);
        // (every word through readfirstlane: the descriptor must sit in scalar registers, or each load becomes a waterfall loop)
        const unsigned long long bp = reinterpret_cast<unsigned long long>(is_a ? g.A : g.B);
        const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bp), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bp >> 32));
        s.rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((unsigned long long)bhi << 32 | blo), (short)0,
                                                 __builtin_amdgcn_readfirstlane((int)(rows * ld * 4)), 0x00020000);
        s.ld4 = __builtin_amdgcn_readfirstlane((unsigned)ld * 4u);
        s.step = TK * s.ld4;
        s.off = ok ? (unsigned)(((kbeg + 4 * kq + shift) * ld + (is_a ? m0 : n0) + c) * 4) : 0x80000000u;
        s.tb = __builtin_amdgcn_readfirstlane(2 * BUF + tap * X6W2_MAX_HW);
        s.isx = !is_a;
        if constexpr (S2) {
            if (!is_a) s.off = ok ? (unsigned)((kbeg + 4 * kq) / HW) * s2_img + (unsigned)(n0 + c) * 4u : 0x80000000u;
            s.tb = __builtin_amdgcn_readfirstlane(2 * BUF + tap * s2_row);
        }
        s.st = gi * GRP + st_off;
        return s;
    };
    Grp G1 = grp_of(wave), G2 = grp_of(wave + 8 < NG ? wave + 8 : wave);
    const bool two = wave + 8 < NG;
    int pk = (kbeg + 4 * kq) % HW;                            // pixel (inside its image) of this lane's first row of the step being loaded

    f32x4 la[4], lb[4];
    unsigned m1 = 0, m2 = 0;                                  // table bytes of the step being loaded
    // row q of the lane's quad for the step whose offsets G holds
    // (S2: `msk` is the table entry of the row -- read some MFMAs earlier -- for a group of X, unused for dY)
    auto issue_row = [&](const Grp& G, unsigned msk, int q, f32x4 (&r)[4]) {
        unsigned o;
        if constexpr (S2) o = G.off + (G.isx ? msk : q * G.ld4);
        else o = (msk >> q) & 1u ? G.off + q * G.ld4 : 0x80000000u;
        r[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(G.rs, (int)o, 0, 0));
    };
    auto entry = [&](const Grp& G, int q) { return *reinterpret_cast<const unsigned*>(lds + G.tb + 4 * (pk + q)); };   // S2
    auto issue = [&](const Grp& G, unsigned msk, int qp, f32x4 (&r)[4]) {
        if constexpr (S2) {
            issue_row(G, entry(G, 2 * qp), 2 * qp, r);
            issue_row(G, entry(G, 2 * qp + 1), 2 * qp + 1, r);
        } else {
            issue_row(G, msk, 2 * qp, r);
            issue_row(G, msk, 2 * qp + 1, r);
        }
    };
    unsigned winc = 0;                                        // S2: what the step just passed added to the image base
    auto next_pixel = [&]() {
        pk += TK;
        if constexpr (S2) winc = pk >= HW ? s2_img : 0u;
        pk = pk >= HW ? pk - HW : pk;
    };
    auto advance = [&](Grp& G) {                              // (after next_pixel)
        if constexpr (S2) G.off += G.isx ? winc : G.step;
        else G.off += G.step;
    };
    // two of the lane's four rows (2 qp, 2 qp + 1), one of its four channels -> 4 bytes in each plane: a quarter-unit, cut into three
    // pieces of ~5 vector instructions (split3_pk of common.hpp, step by step) so that each fits under ONE MFMA of the loop
    struct Quarter { unsigned h, m; float r0, r1; };
    auto q_first = [&](Quarter& s, int qp, int j, const f32x4 (&r)[4]) {
        const float x0 = r[2 * qp][j], x1 = r[2 * qp + 1][j];
        s.h = pk_bf16(x0, x1);
        s.r0 = sub_f32(x0, __uint_as_float(s.h << 16));
        s.r1 = sub_f32(x1, __uint_as_float(s.h & 0xFFFF0000u));
    };
    auto q_second = [&](Quarter& s) {
        s.m = pk_bf16(s.r0, s.r1);
        s.r0 = sub_f32(s.r0, __uint_as_float(s.m << 16));
        s.r1 = sub_f32(s.r1, __uint_as_float(s.m & 0xFFFF0000u));
    };
    auto q_third = [&](const Quarter& s, const Grp& G, int buf, int qp, int j) {
        unsigned char* d = lds + G.st + buf * BUF + 4 * qp + slot_of(cin + j) * 16;
        *reinterpret_cast<unsigned*>(d) = s.h;
        *reinterpret_cast<unsigned*>(d + 2048) = s.m;
        *reinterpret_cast<unsigned*>(d + 4096) = pk_bf16(s.r0, s.r1);
    };
    auto quarter = [&](const Grp& G, int buf, int qp, int j, const f32x4 (&r)[4]) {
        Quarter s;
        q_first(s, qp, j, r);
        q_second(s);
        q_third(s, G, buf, qp, j);
    };

    f32x16 acc[NACC];
#pragma unroll
    for (int tp = 0; tp < NACC; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tp][r] = 0.f;
    const int fa = (wm >> 1) * GRP + kh * 1024 + (wm & 1) * 512 + slot_of(i) * 16;       // + plane * 2048
    const int fb = (NGA + tap0) * GRP + kh * 1024 + wn * 512 + slot_of(i) * 16;           // + tap * GRP + plane * 2048

    __syncthreads();                                          // the table
    // prologue: step 0 -> buffer 0; step 1 in flight
    if constexpr (!S2) { m1 = lds[G1.tb + pk]; m2 = lds[G2.tb + pk]; }
    issue(G1, m1, 0, la); issue(G1, m1, 1, la);
    if (two) { issue(G2, m2, 0, lb); issue(G2, m2, 1, lb); }
    next_pixel(); advance(G1); advance(G2);
#pragma unroll
    for (int qp = 0; qp < 2; ++qp)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            quarter(G1, 0, qp, j, la);
            if (two) quarter(G2, 0, qp, j, lb);
        }
    if constexpr (!S2) { m1 = lds[G1.tb + pk]; m2 = lds[G2.tb + pk]; }
    issue(G1, m1, 0, la); issue(G1, m1, 1, la);
    if (two) { issue(G2, m2, 0, lb); issue(G2, m2, 1, lb); }
    next_pixel(); advance(G1); advance(G2);
    __syncthreads();

    // step t: MFMAs on buffer t & 1; the registers hold step t + 1 (-> buffer (t + 1) & 1), loads for step t + 2 follow each half
    // of a group as soon as its registers are free.  Past the slab's end this splits and stores rows that nobody reads and loads
    // rows of the next slab (or zeros past the tensor): harmless, and it keeps the step free of branches.
    auto run = [&](auto two_c) {
        constexpr bool TWO = decltype(two_c)::value;
        for (int t = 0; t < nk; ++t) {
            const unsigned char* bufp = lds + (t & 1) * BUF;
            const int nb = (t + 1) & 1;
            // fragments: dY's three planes once per step; X's per tap -- plane 0 (first and last product of a tap) double-buffered,
            // planes 2 and 1 re-read into the same registers as soon as their last product of the tap has issued
            uint4 af[3], b0[2], b1, b2;
#pragma unroll
            for (int p = 0; p < 3; ++p) af[p] = *reinterpret_cast<const uint4*>(bufp + fa + p * 2048);
            b0[0] = *reinterpret_cast<const uint4*>(bufp + fb);
            b1 = *reinterpret_cast<const uint4*>(bufp + fb + 2048);
            b2 = *reinterpret_cast<const uint4*>(bufp + fb + 4096);
            const unsigned n1 = S2 ? 0u : lds[G1.tb + pk], n2 = (!S2 && TWO) ? lds[G2.tb + pk] : 0u;          // masks of step t + 2
            unsigned tq[2] = {0u, 0u};                        // S2: the table entries of the row pair about to be re-loaded
#pragma unroll
            for (int tp = 0; tp < NACC; ++tp) {
                if (WGM != 4 && tp >= ntap) break;            // (wave-uniform; WGM = 4: never)
                const int cb = tp & 1;
                const bool more = tp + 1 < NACC;
                const unsigned char* nx = bufp + fb + (tp + 1) * GRP;
                // Work between this tap's MFMAs.  Quarter-unit u = 0 .. 15: group u >> 3, rows qp = (u >> 2) & 1, channel j = u & 3, in three
                // pieces; the fourth quarter of a row pair re-issues that pair's loads (for step t + 2) with its second and third piece,
                // when the rows' registers are free.  WGM = 4 (nine taps): two quarters per tap, one piece per MFMA; WGM = 2 (five or
                // four taps): four quarters per tap, two pieces per MFMA
                constexpr int PER = WGM == 4 ? 2 : 4;
                Quarter qs[PER];
                auto piece = [&](int e) {                     // e = 0 .. 3 PER - 1: piece e % 3 of the tap's quarter e / 3
                    const int u = PER * tp + e / 3, k = e % 3;
                    if (u >= (TWO ? 16 : 8)) return;
                    const int qp = (u >> 2) & 1, j = u & 3;
                    Quarter& s = qs[e / 3];
                    if (u < 8) {
                        if (k == 0) { q_first(s, qp, j, la); if (S2 && j == 2) { tq[0] = entry(G1, 2 * qp); tq[1] = entry(G1, 2 * qp + 1); } }
                        else if (k == 1) { q_second(s); if (j == 3) issue_row(G1, S2 ? tq[0] : n1, 2 * qp, la); }
                        else { q_third(s, G1, nb, qp, j); if (j == 3) issue_row(G1, S2 ? tq[1] : n1, 2 * qp + 1, la); }
                    } else {
                        if (k == 0) { q_first(s, qp, j, lb); if (S2 && j == 2) { tq[0] = entry(G2, 2 * qp); tq[1] = entry(G2, 2 * qp + 1); } }
                        else if (k == 1) { q_second(s); if (j == 3) issue_row(G2, S2 ? tq[0] : n2, 2 * qp, lb); }
                        else { q_third(s, G2, nb, qp, j); if (j == 3) issue_row(G2, S2 ? tq[1] : n2, 2 * qp + 1, lb); }
                    }
                };
                auto slot = [&](int si) {                     // the work after the tap's MFMA si
                    if (WGM == 4) piece(si);
                    else { piece(2 * si); piece(2 * si + 1); }
                };
#define PECLR_FENCE __builtin_amdgcn_sched_barrier(0)
                acc[tp] = mma(af[2], b0[cb], acc[tp]);       PECLR_FENCE;
                if (more) b0[cb ^ 1] = *reinterpret_cast<const uint4*>(nx);
                slot(0);                                      PECLR_FENCE;
                acc[tp] = mma(af[0], b2, acc[tp]);           PECLR_FENCE;
                if (more) b2 = *reinterpret_cast<const uint4*>(nx + 4096);
                slot(1);                                      PECLR_FENCE;
                acc[tp] = mma(af[1], b1, acc[tp]);           PECLR_FENCE;
                slot(2);                                      PECLR_FENCE;
                acc[tp] = mma(af[1], b0[cb], acc[tp]);       PECLR_FENCE;
                slot(3);                                      PECLR_FENCE;
                acc[tp] = mma(af[0], b1, acc[tp]);           PECLR_FENCE;
                if (more) b1 = *reinterpret_cast<const uint4*>(nx + 2048);
                slot(4);                                      PECLR_FENCE;
                acc[tp] = mma(af[0], b0[cb], acc[tp]);       PECLR_FENCE;
                slot(5);                                      PECLR_FENCE;
#undef PECLR_FENCE
            }
            next_pixel();
            advance(G1);
            if (TWO) advance(G2);
            __syncthreads();
        }
    };
    if (two) run(std::true_type{});
    else run(std::false_type{});

    float* out = g.slabs + (size_t)blockIdx.y * g.M * g.ldc;
    float* wlds = reinterpret_cast<float*>(lds) + wave * (32 * XEPL);
    const int er = lane >> 3, ec = (lane & 7) * 4;
    const int mt = m0 + wm * 32, nt = n0 + wn * 32;
#pragma unroll
    for (int tp = 0; tp < NACC; ++tp) {
        if (WGM != 4 && tp >= ntap) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = acc[tp][r];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = mt + er + 8 * jj, n = nt + ec;
            const float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
            if (m < g.M && n < g.N) *reinterpret_cast<float4*>(out + (size_t)m * g.ldc + (tap0 + tp) * g.N + n) = c;
        }
    }
}

// ---- gemm_x6t_kernel (1x1 weight gradients, GEO = false) with its loop re-scheduled the same way (round 6): same tiles, same LDS
// layout, same order of MFMAs per accumulator (bit-identical slabs).  Every wave owns one 64-column group of the split (or none);
// its eight quarter-units, three pieces each, follow the step's 6 MT NT MFMAs at even distances; rows arrive by buffer loads
// (out of range = zeros; the non-temporal hint of rows only one workgroup reads is kept -- a wave-uniform choice between two
// load instructions); the B fragments of the second half of the output columns are fetched under the MFMAs of the first.
// Measured at ResNet-50's shapes (tools/exp/x6w_ablate.hip 1, profiles/r06_x6t_ablate.txt): 5 - 28 % faster on every tile EXCEPT
// 256 x 256 (12 - 26 % slower in every variant tried: fenced / unfenced, 24 pieces / four lumps) -- there the 48 MFMAs of a step
// write eight independent accumulators, hipcc's "all MFMAs, then all vector work" already keeps the matrix cores fed from the
// SIMD's other wave, and pieces between them only add issue stalls.  So the library keeps the first form for that tile.
// PF2: two register sets for the rows -- loads run TWO k-steps ahead of the split (the loop unrolled by two: a set holds the steps
// of one parity).  For the 64-wide tiles (layer1's gradients: HBM-bound, four to eight MFMAs per wave and step, so that one step
// does not cover a round trip to HBM): 122 / 217 / 220 us against 135 / 237 / 242 with one set, 137 / 238 / 246 in the first form
// (profiles/r06_x6t_ablate_pf2.txt); no gain, or a loss, on the wider tiles, which keep one set.
template <int MT, int NT, int WGM, bool PF2 = false>
__global__ __launch_bounds__(512, 2) void gemm_x6t2_kernel(X6TArgs g) {
    constexpr int WGN = 8 / WGM;
    constexpr int TM = 32 * WGM * MT, TN = 32 * WGN * NT;
    static_assert((TM + TN) / 64 <= 8 && TM % 64 == 0 && TN % 64 == 0, "one 64-column group of the split per wave");
    constexpr int HALF_A = TM * 16, HALF_B = TN * 16;         // bytes of one k-half of a plane
    constexpr int PL_A = 2 * HALF_A, PL_B = 2 * HALF_B;
    constexpr int BUF = 3 * (PL_A + PL_B);                    // one k-step of both operands
    constexpr int XEPL = 36;
    static_assert(2 * BUF >= 8 * 32 * XEPL * 4, "epilogue transposes live in the plane buffers");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int i = lane & 31, kh = lane >> 5;
    const int nct = (g.N + TN - 1) / TN;
    const int m0 = (int)(blockIdx.x / nct) * TM, n0 = (int)(blockIdx.x % nct) * TN;
    const int kbeg = blockIdx.y * g.kchunk, kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + TK - 1) / TK;

    // this wave's group of the split (gemm_x6t_kernel's roles)
    const bool is_a = wave < TM / 64;
    const bool active = wave < (TM + TN) / 64;
    const int cgrp = is_a ? wave : wave - TM / 64;
    const int chunk = (lane & 7) + 8 * (lane >> 5);           // 0..15 within the 64-column group
    const int kq = (lane >> 3) & 3;
    const int col = cgrp * 64 + 4 * chunk;                    // first of this thread's 4 columns inside the tile
    const int ld = is_a ? g.lda : g.ldb;
    const bool col_ok = active && (is_a ? m0 + col < g.M : n0 + col < g.N);
    const int tile = col >> 5, cin = col & 31;
    const int st_base = (kq >> 1) * (is_a ? HALF_A : HALF_B) + tile * 512 + (kq & 1) * 8 + (is_a ? 0 : 3 * PL_A);
    const int st_plane = is_a ? PL_A : PL_B;
    const bool once = is_a ? nct == 1 : (int)gridDim.x == nct;     // rows exactly one workgroup of a split reads: non-temporal

    const unsigned long long bp = reinterpret_cast<unsigned long long>(is_a ? g.A : g.B);
    const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)bp), bhi = __builtin_amdgcn_readfirstlane((unsigned)(bp >> 32));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((unsigned long long)bhi << 32 | blo), (short)0,
                                                                        __builtin_amdgcn_readfirstlane(g.K * ld * 4), 0x00020000);
    const unsigned ld4b = __builtin_amdgcn_readfirstlane((unsigned)ld * 4u), stepb = TK * ld4b;
    unsigned off = col_ok ? (unsigned)(((kbeg + 4 * kq) * ld + (is_a ? m0 : n0) + col) * 4) : 0x80000000u;

    f32x4 r4s[PF2 ? 2 : 1][4];
    auto issue_row = [&](int q, int set = 0) {                // row q of the lane's quad, step = where `off` stands
        if (once) r4s[set][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + q * ld4b), 0, 2));
        else r4s[set][q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off + q * ld4b), 0, 0));
    };
    struct Quarter { unsigned h, m; float r0, r1; };
    auto q_first = [&](Quarter& s, int qp, int j, int set = 0) {
        const float x0 = r4s[set][2 * qp][j], x1 = r4s[set][2 * qp + 1][j];
        s.h = pk_bf16(x0, x1);
        s.r0 = sub_f32(x0, __uint_as_float(s.h << 16));
        s.r1 = sub_f32(x1, __uint_as_float(s.h & 0xFFFF0000u));
    };
    auto q_second = [&](Quarter& s) {
        s.m = pk_bf16(s.r0, s.r1);
        s.r0 = sub_f32(s.r0, __uint_as_float(s.m << 16));
        s.r1 = sub_f32(s.r1, __uint_as_float(s.m & 0xFFFF0000u));
    };
    auto q_third = [&](const Quarter& s, int buf, int qp, int j) {
        unsigned char* d = lds + buf * BUF + st_base + 4 * qp + slot_of(cin + j) * 16;
        *reinterpret_cast<unsigned*>(d) = s.h;
        *reinterpret_cast<unsigned*>(d + st_plane) = s.m;
        *reinterpret_cast<unsigned*>(d + 2 * st_plane) = pk_bf16(s.r0, s.r1);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][y][r] = 0.f;

    const int fa = kh * HALF_A + (wm * MT) * 512 + slot_of(i) * 16;                  // + a * 512 + plane * PL_A
    const int fb = 3 * PL_A + kh * HALF_B + (wn * NT) * 512 + slot_of(i) * 16;       // + y * 512 + plane * PL_B

    // prologue: step 0 -> buffer 0; step 1 in flight.  (Rows past the tensor read as zeros, rows past this slab are never multiplied.)
    if (active) {
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_row(q);
        off += stepb;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            Quarter s;
            q_first(s, u >> 2, u & 3);
            q_second(s);
            q_third(s, 0, u >> 2, u & 3);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) issue_row(q, PF2 ? 1 : 0);
        off += stepb;
        if constexpr (PF2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) issue_row(q, 0);                              // step 2
            off += stepb;
        }
    }
    __syncthreads();

    constexpr int NH = NT > 2 ? 2 : 1, YH = NT / NH;          // B fragments in two halves (registers)
    constexpr int NM = 6 * MT * NT;                           // MFMAs per step; piece e of 24 follows MFMA e NM / 24
    auto run = [&](auto active_c) {
        constexpr bool ACT = decltype(active_c)::value;
        for (int t0 = 0; t0 < nk; t0 += (PF2 ? 2 : 1))
#pragma unroll
        for (int par = 0; par < (PF2 ? 2 : 1); ++par) {
            const int t = t0 + par;
            if (PF2 && t >= nk) break;                        // (uniform)
            const int set = PF2 ? (par ^ 1) : 0;              // the set that holds step t + 1
            const unsigned char* bufp = lds + (t & 1) * BUF;
            const int nb = (t + 1) & 1;
            uint4 af[MT][3], b0[2][YH], b1[YH], b2[YH];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int p = 0; p < 3; ++p) af[a][p] = *reinterpret_cast<const uint4*>(bufp + fa + a * 512 + p * PL_A);
#pragma unroll
            for (int y = 0; y < YH; ++y) {
                b0[0][y] = *reinterpret_cast<const uint4*>(bufp + fb + y * 512);
                b1[y] = *reinterpret_cast<const uint4*>(bufp + fb + y * 512 + PL_B);
                b2[y] = *reinterpret_cast<const uint4*>(bufp + fb + y * 512 + 2 * PL_B);
            }
            Quarter qs;
            auto work_after = [&](int n) {                    // the pieces that follow MFMA n of the step
                if (!ACT) return;
#pragma unroll
                for (int e = 0; e < 24; ++e) {
                    if (e * NM / 24 != n) continue;
                    const int u = e / 3, k = e % 3, qp = u >> 2, j = u & 3;
                    if (k == 0) q_first(qs, qp, j, set);
                    else if (k == 1) { q_second(qs); if (j == 3) issue_row(2 * qp, set); }
                    else { q_third(qs, nb, qp, j); if (j == 3) issue_row(2 * qp + 1, set); }
                }
            };
#define PECLR_FENCE __builtin_amdgcn_sched_barrier(0)
#pragma unroll
            for (int half = 0; half < NH; ++half) {
                const bool more = half + 1 < NH;
                const unsigned char* nx = bufp + fb + (half + 1) * YH * 512;
#pragma unroll
                for (int pr = 0; pr < 6; ++pr) {
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int y = 0; y < YH; ++y)
#pragma unroll
                        for (int a = 0; a < MT; ++a) {
                            const uint4& bb = QB[pr] == 0 ? b0[half & 1][y] : (QB[pr] == 1 ? b1[y] : b2[y]);
                            acc[a][half * YH + y] = mma(af[a][PA[pr]], bb, acc[a][half * YH + y]);
                            PECLR_FENCE;
                            const int n = (half * 6 + pr) * YH * MT + y * MT + a;
                            // fragments of the second half: plane 0 into its second register set at once, planes 2 / 1 into the
                            // registers their last product of this half has just left
                            if (more && pr == 0 && a == MT - 1) b0[(half + 1) & 1][y] = *reinterpret_cast<const uint4*>(nx + y * 512);
                            if (more && pr == 1 && a == MT - 1 && y == YH - 1) {
#pragma unroll
                                for (int yy = 0; yy < YH; ++yy) b2[yy] = *reinterpret_cast<const uint4*>(nx + yy * 512 + 2 * PL_B);
                            }
                            if (more && pr == 4 && a == MT - 1 && y == YH - 1) {
#pragma unroll
                                for (int yy = 0; yy < YH; ++yy) b1[yy] = *reinterpret_cast<const uint4*>(nx + yy * 512 + PL_B);
                            }
                            work_after(n);
                            PECLR_FENCE;
                        }
                }
            }
#undef PECLR_FENCE
            if (ACT) off += stepb;
            __syncthreads();
        }
    };
    if (active) run(std::true_type{});
    else run(std::false_type{});

    // epilogue: wave-private 32 x 32 transposes through LDS, 16-byte stores into this split's slab
    float* out = g.slabs + (size_t)blockIdx.y * g.M * g.ldc;
    float* wlds = reinterpret_cast<float*>(lds) + wave * (32 * XEPL);
    const int er = lane >> 3, ec = (lane & 7) * 4;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int y = 0; y < NT; ++y) {
            const int mt = m0 + (wm * MT + a) * 32, nt = n0 + (wn * NT + y) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = acc[a][y][r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = mt + er + 8 * jj, n = nt + ec;
                const float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
                if (m < g.M && n < g.N) *reinterpret_cast<float4*>(out + (size_t)m * g.ldc + n) = c;
            }
        }
}

constexpr int PF_SKINNY = 0;                                 // (tag of the 64-wide tiles in the launch table below; the first form's PF is 1 for every tile)
struct TilePick { int mt, nt, wgm; };
// 1x1: workgroup tile 32 wgm mt x 256 / wgm nt.  64-wide sides (layer1) get tiles that are 64 wide on that side.
inline TilePick pick_tile(int M, int N) {
    if (M <= 64) return {1, N >= 256 ? 2 : 1, 2};                  // 64 x 256 / 64 x 128
    if (N <= 64) return {M >= 256 ? 2 : 1, 1, 4};                  // 256 x 64 / 128 x 64
    return {M >= 256 ? 2 : 1, N >= 256 ? 4 : 2, 4};                // 256 x 256 / 256 x 128 / 128 x 256 / 128 x 128
}
inline int tile_m(const TilePick& t) { return 32 * t.wgm * t.mt; }
inline int tile_n(const TilePick& t) { return 32 * (8 / t.wgm) * t.nt; }

}  // namespace
}  // namespace peclr

using namespace peclr;

// Number of K splits (= slabs): about one workgroup (8 waves) per CU -- two where the planes of two fit the LDS (the
// HBM-bound 64-wide gradients) --, at least 16 k-steps per workgroup.
extern "C" int peclr_gemm_x6t_slabs(int M, int N, int K, int taps) {
    if (M <= 0 || N <= 0 || K <= 0 || (taps != 1 && taps != 9)) return 0;
    const TilePick t = pick_tile(M, N);
    const int tm = taps == 9 ? (M <= 64 ? 64 : 128) : tile_m(t), tn = taps == 9 ? 64 : tile_n(t);    // 3x3: all nine taps in one workgroup
    const long tiles = (long)((M + tm - 1) / tm) * ((N + tn - 1) / tn);
    const long wgs = taps == 1 && tm + tn <= 320 ? 512 : 256;
    long s = (wgs + tiles - 1) / tiles;
    const long max_s = (K + 16 * TK - 1) / (16 * TK);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const int kchunk = (int)(((K + s - 1) / s + TK - 1) / TK * TK);
    return (K + kchunk - 1) / kchunk;
}

// stride 2: A's K rows are the H x W output pixels of a stride-2 convolution (1x1 without padding, or 3x3 with padding 1),
// B's 4 K rows the 2H x 2W input pixels.
static int gemm_x6t_host(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* slabs,
                         int n_slabs, int taps, int H, int W, int stride, const float* zeros, peclr_stream_t stream) {
    if (!A || !B || !slabs || !zeros) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || n_slabs < 1 || (taps != 1 && taps != 9) || (stride != 1 && stride != 2)) return PECLR_ERR_SHAPE;
    if (M % 4 || N % 4 || lda % 4 || ldb % 4 || lda < M || ldb < N) return PECLR_ERR_SHAPE;
    if ((taps == 9 || stride == 2) && (H <= 0 || W < 6 || K % (H * W))) return PECLR_ERR_SHAPE;
    if (stride == 2 && (long)K * 4 > 0x7fffffffL) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(B) || !aligned16(slabs) || !aligned16(zeros)) return PECLR_ERR_ALIGN;
    if (n_slabs != peclr_gemm_x6t_slabs(M, N, K, taps)) return PECLR_ERR_WORKSPACE;
    X6TArgs g;
    g.A = A; g.B = B; g.slabs = slabs;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = taps * N;
    g.kchunk = ((K + n_slabs - 1) / n_slabs + TK - 1) / TK * TK;
    g.taps = taps; g.H = H; g.W = W; g.stride = stride; g.zeros = zeros;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (taps == 9) {
        // the re-scheduled loop (gemm_x6w2_kernel): stride 1, images of at most 56 x 56 output pixels (its table), operands below 2 GiB
        // (32-bit buffer offsets); PECLR_X6W2=0 keeps the first form for A/B runs
        static const bool x6w2 = getenv("PECLR_X6W2") ? atoi(getenv("PECLR_X6W2")) != 0 : true;
        if (x6w2 && stride == 1 && H * W <= X6W2_MAX_HW && H * W >= TK && K > 2 * (W + 2) && (long)(K + W + 2) * (lda > ldb ? lda : ldb) * 4 < 0x7fffffffL) {
            if (M <= 64) hipLaunchKernelGGL(gemm_x6w2_kernel<2>, dim3(((M + 63) / 64) * ((N + 63) / 64), n_slabs), dim3(512), 0, s, g);
            else hipLaunchKernelGGL(gemm_x6w2_kernel<4>, dim3(((M + 127) / 128) * ((N + 63) / 64), n_slabs), dim3(512), 0, s, g);
            return launch_status();
        }
        // ... its stride-2 arm: output images of at most 28 x 28 pixels (four table bytes per tap and pixel), X below 1 GiB (2^30 marks "no row")
        if (x6w2 && stride == 2 && H * W <= X6W2_S2_MAX_HW && H * W >= TK && (long)(K + 2 * TK) * lda * 4 < 0x7fffffffL &&
            4L * (K + 2 * H * W) * ldb * 4 < 0x3fffffffL) {
            if (M <= 64) hipLaunchKernelGGL((gemm_x6w2_kernel<2, true>), dim3(((M + 63) / 64) * ((N + 63) / 64), n_slabs), dim3(512), 0, s, g);
            else hipLaunchKernelGGL((gemm_x6w2_kernel<4, true>), dim3(((M + 127) / 128) * ((N + 63) / 64), n_slabs), dim3(512), 0, s, g);
            return launch_status();
        }
        if (M <= 64) hipLaunchKernelGGL(gemm_x6w_kernel<2>, dim3(((M + 63) / 64) * ((N + 63) / 64), n_slabs), dim3(512), 0, s, g);
        else hipLaunchKernelGGL(gemm_x6w_kernel<4>, dim3(((M + 127) / 128) * ((N + 63) / 64), n_slabs), dim3(512), 0, s, g);
        return launch_status();
    }
    const TilePick t = pick_tile(M, N);
    const dim3 grid(((M + tile_m(t) - 1) / tile_m(t)) * ((N + tile_n(t) - 1) / tile_n(t)), n_slabs);
    // the re-scheduled loop (gemm_x6t2_kernel): stride 1, operands below 2 GiB (32-bit buffer offsets); PECLR_X6T2=0 keeps the first form
    static const bool x6t2_on = getenv("PECLR_X6T2") ? atoi(getenv("PECLR_X6T2")) != 0 : true;
    const bool x6t2 = x6t2_on && stride == 1 && (long)(K + 2 * TK) * (lda > ldb ? lda : ldb) * 4 < 0x7fffffffL;
#define PECLR_LAUNCH(MT_, NT_, WGM_, PF_)                                                                            \
    do {                                                                                                             \
        if (stride == 2) hipLaunchKernelGGL((gemm_x6t_kernel<MT_, NT_, WGM_, true, 1>), grid, dim3(512), 0, s, g);   \
        else if (x6t2) hipLaunchKernelGGL((gemm_x6t2_kernel<MT_, NT_, WGM_, (PF_) == PF_SKINNY>), grid, dim3(512), 0, s, g); \
        else hipLaunchKernelGGL((gemm_x6t_kernel<MT_, NT_, WGM_, false, 1>), grid, dim3(512), 0, s, g);              \
    } while (0)
    if (t.wgm == 2) {
        if (t.nt == 2) PECLR_LAUNCH(1, 2, 2, PF_SKINNY);
        else PECLR_LAUNCH(1, 1, 2, PF_SKINNY);
    } else if (t.nt == 1) {
        if (t.mt == 2) PECLR_LAUNCH(2, 1, 4, PF_SKINNY);
        else PECLR_LAUNCH(1, 1, 4, PF_SKINNY);
    } else if (t.mt == 2 && t.nt == 4) {                       // 256 x 256: the first form is the faster one (see gemm_x6t2_kernel)
        if (stride == 2) hipLaunchKernelGGL((gemm_x6t_kernel<2, 4, 4, true, 1>), grid, dim3(512), 0, s, g);
        else hipLaunchKernelGGL((gemm_x6t_kernel<2, 4, 4, false, 1>), grid, dim3(512), 0, s, g);
    }
    else if (t.mt == 2) PECLR_LAUNCH(2, 2, 4, 1);
    else if (t.nt == 4) PECLR_LAUNCH(1, 4, 4, 1);
    else PECLR_LAUNCH(1, 2, 4, 1);
#undef PECLR_LAUNCH
    return launch_status();
}

extern "C" int peclr_gemm_x6t_f32(int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* slabs,
                                  int n_slabs, int taps, int H, int W, int stride, const float* zeros, peclr_stream_t stream) {
    return gemm_x6t_host(M, N, K, A, lda, B, ldb, slabs, n_slabs, taps, H, W, stride, zeros, stream);
}
