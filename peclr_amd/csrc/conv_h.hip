// 16-bit convolutions of the residual blocks (bf16 / fp16 autocast backbones: BASELINE configs C3 and C5, and the
// reference's own `precision: 16`, training_config.json:9) as GEMMs on v_mfma_f32_32x32x16_{bf16,f16} with fp32 accumulation:
//   C[M,N] (half) = A[M,K] (half) . W[N,K]^T (+ addend),   1x1 / 3x3, stride 1 / 2, forward and input gradient,
// with the neighbouring BatchNorm's statistics (forward) or backward reduction (input gradient) in the epilogue.
//
// The operands are already 16-bit, so -- unlike the fp32 six-product kernels (gemm_x6p.hip), whose tile structure, XCD-aware
// tile order, partial-sum layouts and epilogue variants this file shares -- nothing is split: the main loop is fragment
// reads, MFMAs and one raw barrier per 32 k, and BOTH operands arrive by LDS-DMA (no VGPRs, no VALU, no ds_write):
//   * W: packed once per optimiser step from the fp32 MASTER weights (peclr_h_pack: the cast autocast performs per forward
//     rides in the same launch) into MFMA fragment order -- per (128 output columns, 32 k) one contiguous 8 KiB chunk of
//     eight 1 KiB pieces [32-column block][k-half of 16], a piece being lane l's 16 bytes at l * 16 = column n0 + (l & 31),
//     k = k0 + 8 (l >> 5) .. + 7;
//   * A: every wave owns 32 * WM rows x all output columns of the workgroup tile, and brings ITS rows in itself: one
//     `global_load_lds_dwordx4` moves 16 rows x 64 bytes (32 k), four lanes per row, into a wave-private [row][4 x 16 B]
//     image whose 16-byte slots are XOR-swizzled by (row >> 2) & 3 -- the DMA writes lane-linear, so the swizzle is applied
//     to the GLOBAL address each lane fetches -- which makes the MFMA fragment reads (ds_read_b128, row pitch 64 B)
//     bank-conflict free.
// These GEMMs are bound by HBM and by operand traffic from L2, not by the matrix cores (26 GFLOP per 1x1 product is 10 us
// at the bf16 peak; its 130 - 510 MB are 20 - 80 us at the achievable 6.3 TB/s): three stages of 24 KiB in flight per
// workgroup, two workgroups per CU, 256-row tiles so that A is re-read from L2 once per 128 output columns only.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

#include <cstddef>

namespace peclr {
namespace {

typedef uint16_t h16_t;                  // storage of both 16-bit formats
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

constexpr int HN = 128;                  // output columns per packed chunk
constexpr int HK = 32;                   // k per step (two MFMA k-extents)
constexpr int HCHUNK = 8 * 1024;         // bytes of packed W per (128 columns, 32 k)

struct BF16 {
    static constexpr int io = PECLR_DTYPE_BF16;
    static __device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float up(unsigned lo16) { return __uint_as_float(lo16 << 16); }
    static __device__ __forceinline__ float lo(unsigned w) { return __uint_as_float(w << 16); }           // the two halves of a word,
    static __device__ __forceinline__ float hi(unsigned w) { return __uint_as_float(w & 0xFFFF0000u); }   // one instruction each
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pk_bf16(a, b); }          // round to nearest even
};
struct F16 {
    static constexpr int io = PECLR_DTYPE_F16;
    static __device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ float up(unsigned lo16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)lo16); }
    static __device__ __forceinline__ float lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)w); }
    static __device__ __forceinline__ float hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        const f32x2_t v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
    }
};

struct HArgs {
    const h16_t* A;
    const void* Bp;                      // packed W (peclr_h_pack)
    const h16_t* addend;
    h16_t* out;
    int M, N, K, lda, ldo, ldd;
    int add_h, add_w;                    // > 0: the addend holds every second pixel of add_h x add_w images (compact stride-2 gradient)
    const unsigned* add_mask;            // optional 1-bit mask of the addend ([M][N / 32])
    const float* stat_shift;             // optional BatchNorm statistics of the output (layout of peclr_bn2d_stats' partials)
    float* stat_partial;
    int H, W, flip;                      // TAPS = 9: rows are the pixels of H x W images; flip = the input gradient's filter
    int stride, Hin, Win;                // stride = 2 (forward): rows are OUTPUT pixels of an Hin x Win input
    int s2d;                             // TAPS = 9: input gradient of the stride-2 3x3, one parity class per blockIdx.y
    int Mp;                              // RING: pixels of the padded space, images x (H + 1) x (W + 1)
    const h16_t* zeros;                  // >= 64 bytes of zeros (padding pixels)
    // optional: C is the gradient arriving at a BatchNorm2d(+ReLU) layer -> its backward reduction in the epilogue
    const h16_t* bb_x;
    const float* bb_mean;
    const float* bb_invstd;
    const float* bb_ss;
    const unsigned* bb_mask;
    int bb_relu;
    float* bb_partial;
    int stream_out;                      // the output (and its addend) is larger than the caches: non-temporal epilogue
};

// 16 bytes per lane, global -> LDS at (wave-uniform) dst + lane * 16 (inline assembly: see gemm_x6p.hip -- through the
// builtin hipcc makes every later ds_read wait for the DMA; here every wait on the vm counter is written by hand)
__device__ __forceinline__ void hdma16(const void* src, unsigned lds_byte_offset) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds_byte_offset) : "memory", "m0");
}
// A/B builds: bit 0 = the activation rows of 1x1 products with ONE column tile are fetched with the non-temporal hint.  Measured
// slower (round 5, same box, bf16: conv1x1_dgrad 74 -> 82 us, conv1x1_fwd 70 -> 72): off
#ifndef PECLR_CONVH_NT
#define PECLR_CONVH_NT 0
#endif
__device__ __forceinline__ void hdma16_nt(const void* src, unsigned lds_byte_offset) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt"
                 :: "v"(src), "s"(lds_byte_offset) : "memory", "m0");
}

// ... the same with a wave-uniform 64-bit base (scalar registers) + a 32-bit lane offset + an immediate: no vector arithmetic
// per request.  The instruction's immediate offset is added to the global address AND to the LDS address (M0 + offset +
// lane * 16): piece k of a contiguous run is (same base, same M0, offset k * 1024)
template <int IMM>
__device__ __forceinline__ void hdma16s(const void* sbase, unsigned lane_off, unsigned lds_byte_offset) {
    // (s_nop 4: the base may come straight from scalar arithmetic -- a vector memory instruction reading a scalar register the
    // scalar unit has just written needs five wait states, and the compiler does not see into this string)
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :: "v"(lane_off), "s"(sbase), "s"(lds_byte_offset), "n"(IMM) : "memory", "m0");
}

// WM: 32-row MFMA tiles per wave (2 -> 256-row workgroup tile, 1 -> 128); NTL: 32-column tiles per wave (4 -> 128 output
// columns per workgroup, 2 -> 64); TAPS = 9: 3x3 / padding 1 as an implicit GEMM, K ordered (tap, channel)
// EP: epilogue features compiled in (registers are allotted for the largest path of an instantiation, and these launches live
// on workgroups per CU): bit 0 = an addend (dense / compact stride-2 / 1-bit-masked), bit 1 = the BatchNorm backward reduction
#ifdef PECLR_CONV_H_TIMING                                 // experiment builds only (tools/exp/ab): shader clocks per kernel phase
__device__ unsigned long long conv_h_timing[8];
#define PECLR_PHASE(k) do { if (threadIdx.x == 0 && blockIdx.x == 8) { const unsigned long long now_ = __builtin_readcyclecounter(); atomicAdd(&conv_h_timing[k], now_ - tphase); tphase = now_; } } while (0)
#else
#define PECLR_PHASE(k) do { } while (0)
#endif

// RING > 0 (3x3, stride 1, 256-row tiles): the rows of a tile are RING-free consecutive pixels of the PADDED space -- images of
// (H + 1) x (W + 1) pixels whose last row and column are zeros, so that the tap (dh, dw) of pixel p is pixel p + dh (W + 1) + dw
// for every p (the zero column is the right neighbour of a row and the left neighbour of the next, the zero row the bottom of
// an image and the top of the next).  Per 32-channel chunk the workgroup fetches the 256 + 2 (W + 2) pixels its nine taps read
// ONCE (RING = rows of that stage: 320 or 384, W <= 62) and the nine products take their A fragments from shifted rows of it;
// only the W chunks (8 KiB) arrive per step.  Without it every tap fetches its own 256 x 64 B rows: nine times the LDS fill,
// in half-line pieces -- the 3x3 launches ran at 0.19 - 0.26 of the matrix cores (93 - 127 us at every layer of ResNet-50).
// Outputs at padding positions are not stored and not counted in the fused sums (7 % / 13 % / 23 % more rows at 28 / 14 / 7).
template <typename H, int WM, int TAPS, int NTL, int EP, int RING = 0>
__global__ __launch_bounds__(256, (WM == 1 && RING == 0) ? ((EP & 2) ? 3 : 4) : 2) void conv_h_kernel(HArgs g) {
    static_assert(RING == 0 || (TAPS == 9 && RING % 64 == 0 && (WM == 2 || (WM == 1 && RING == 384))),
                  "ring stages: 3x3 on 256-row tiles (W <= 62) or, for rows of 63 ... 126 pixels, on 128-row tiles");
    constexpr bool ADD = (EP & 1) != 0, BBF = (EP & 2) != 0;
    constexpr int RM = 32 * WM, TM = 4 * RM;
    constexpr int NA = RING ? RING / 64 : RM / 16;       // A DMA instructions per wave and k-step / ring stage (16 rows x 64 B each)
    constexpr int NBD = NTL == 4 ? 2 : 1;                // B DMA instructions per wave and k-step
    // LDS-DMA targets must lie below 64 KiB (M0 carries a 16-bit LDS address): three stages of A rows (two k-steps of
    // run-ahead: they come from HBM) and two of W (one step of run-ahead: L2) are 3 x 16 + 2 x 8 = 64 KiB at 256-row tiles.
    // RING: two stages of RING rows (a chunk lasts nine steps) and as many W stages as fit, at most four
    constexpr int BSZ = NTL * 2048;                      // bytes of W per stage
    constexpr int NSA = RING ? 2 : 3;
    constexpr int ASZ = RING ? RING * 64 : TM * 64;      // bytes of activation rows per stage
    constexpr int NSB = !RING ? 2 : ((65536 - NSA * ASZ) / BSZ < 4 ? (65536 - NSA * ASZ) / BSZ : 4);
    static_assert(NSB >= 2, "two W stages at least");
    constexpr int B0 = NSA * ASZ;
    constexpr int OPS = B0 + NSB * BSZ;
    static_assert(OPS <= 65536, "LDS-DMA targets below 64 KiB");
    constexpr int XE = 68;                               // floats per row of the epilogue's 32 x 64 transpose buffer
    constexpr int CST = 4 * 32 * XE * 4 + 4 * 2 * 128 * 4;      // BB: [4][128] per-column constants of the BatchNorm layer
    constexpr int EPI = CST + (BBF ? 4 * 128 * 4 : 0);
    __shared__ __attribute__((aligned(1024))) unsigned char lds[OPS > EPI ? OPS : EPI];   // (the epilogue re-uses the stages)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
#ifdef PECLR_CONV_H_TIMING
    unsigned long long tphase = __builtin_readcyclecounter();
#endif
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)(lptr_t)lds);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);

    constexpr int PNL = 32 * NTL;
    const int nct = (g.N + PNL - 1) / PNL;
    const int j = blockIdx.x / 8;
    const int row_block = 8 * (j / nct) + (int)(blockIdx.x % 8);      // all column tiles of a row block on one XCD
    const int Wp = g.W + 1, Hp = g.H + 1;                             // (RING) the padded image
    if (row_block * TM >= (RING ? g.Mp : g.M)) return;
    auto ring_row = [&](int p) -> int {                               // padded pixel -> row of the NHWC tensor, -1: a zero
        if (p < 0 || p >= g.Mp) return -1;
        const int w = p % Wp, q = p / Wp, h = q % Hp, img = q / Hp;
        return w < g.W && h < g.H ? (img * g.H + h) * g.W + w : -1;
    };
    const int m0 = row_block * TM + wave * RM, ct = j % nct, n0 = ct * PNL;
    const bool s2d = TAPS == 9 && g.s2d;
    const int ph = s2d ? 1 - (int)(blockIdx.y >> 1) : 0, pw = s2d ? 1 - (int)(blockIdx.y & 1) : 0;
    const int ntap = s2d ? (1 + ph) * (1 + pw) : TAPS;
    const int kpt = g.lda / HK;                                       // k-steps per tap (TAPS = 9) / of the whole product
    const int nk = TAPS == 9 ? ntap * kpt : g.K / HK;
    const int nk_all = g.K / HK;
    const unsigned char* bsrc = static_cast<const unsigned char*>(g.Bp) + (size_t)(NTL == 4 ? ct : ct >> 1) * nk_all * HCHUNK +
                                (NTL == 4 ? 0 : (ct & 1) * 4) * 1024 + lane * 16;

    f32x16 acc[WM][NTL];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int y = 0; y < NTL; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][y][r] = 0.f;

    // this lane's A source: row 16 c + (lane >> 2) of the wave's block; it fetches the 16-byte chunk that belongs in LDS slot
    // lane & 3 of that row: chunk = slot ^ ((row >> 2) & 3), and (row >> 2) & 3 == (lane >> 4) & 3 for every c
    const int achunk = (lane & 3) ^ ((lane >> 4) & 3);
    const h16_t* asrc[NA];
    unsigned tapmask[NA];
#pragma unroll
    for (int c = 0; c < NA; ++c) {
        if constexpr (RING != 0) {                        // stage row r = pixel P0 - (W + 2) + r of the padded space
            const int real = ring_row(row_block * TM - (Wp + 1) + 16 * (wave * NA + c) + (lane >> 2));
            asrc[c] = real >= 0 ? g.A + (size_t)real * g.lda + 8 * achunk : nullptr;
            tapmask[c] = 0;
            continue;
        }
        int row = m0 + 16 * c + (lane >> 2);
        row = row < g.M ? row : g.M - 1;
        asrc[c] = g.A + (size_t)row * g.lda + 8 * achunk;
        tapmask[c] = 0;
        if (TAPS == 9 || g.stride == 2) {
            const int ow = row % g.W, oh = (row / g.W) % g.H, img = row / (g.W * g.H);
            const int ih = g.stride * oh, iw = g.stride * ow;
            if (g.stride == 2) asrc[c] = g.A + ((size_t)(img * g.Hin + ih) * g.Win + iw) * g.lda + 8 * achunk;
            if constexpr (TAPS == 9) {
                if (s2d) {                                             // class tap u = ua (1 + pw) + ub reads dY (oh + da, ow + db)
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ua = u / (1 + pw), ub = u - ua * (1 + pw);
                        const int da = ph && ua == 0, db = pw && ub == 0;
                        if (u < ntap && oh + da < g.H && ow + db < g.W) tapmask[c] |= 1u << u;
                    }
                } else {
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        const int dh = g.flip ? 1 - tap / 3 : tap / 3 - 1, dw = g.flip ? 1 - tap % 3 : tap % 3 - 1;
                        if ((unsigned)(ih + dh) < (unsigned)g.Hin && (unsigned)(iw + dw) < (unsigned)g.Win) tapmask[c] |= 1u << tap;
                    }
                }
            }
        }
    }
    const h16_t* const zsrc = TAPS == 9 ? g.zeros + 8 * achunk : nullptr;
    // t / kpt without a division in the loop (t kpt < 2^32: exact)
    const unsigned kmagic = 0xFFFFFFFFu / (unsigned)kpt + 1u;
    auto class_tap = [&](int t, int& rem, int& ftap, int& da, int& db) {      // s2d: k-step t = step `rem` of filter tap `ftap`
        const int u = (int)__umulhi((unsigned)t, kmagic);
        rem = t - u * kpt;
        const int ua = u / (1 + pw), ub = u - ua * (1 + pw);
        da = ph && ua == 0;
        db = pw && ub == 0;
        ftap = 3 * (ph ? 2 * ua : 1) + (pw ? 2 * ub : 1);
    };
    auto issue_b = [&](int t, int step) {                 // this wave's pieces of W chunk t -> W stage step % NSB
        int bt_step = t;
        if constexpr (TAPS == 9) {
            if (s2d) {
                int rem, ftap, da, db;
                class_tap(t, rem, ftap, da, db);
                bt_step = ftap * kpt + rem;
            }
        }
        const unsigned char* s = bsrc + (size_t)bt_step * HCHUNK;
        const unsigned d = lds0 + B0 + (step % NSB) * BSZ;
        if constexpr (NTL == 4) {
            hdma16(s + (2 * wave_s) * 1024, d + (2 * wave_s) * 1024);
            hdma16(s + (2 * wave_s + 1) * 1024, d + (2 * wave_s + 1) * 1024);
        } else {
            hdma16(s + wave_s * 1024, d + wave_s * 1024);
        }
    };
    auto issue_ring = [&](int chunk) {                    // RING: this wave's rows of the 32-channel chunk -> stage chunk & 1
        const unsigned st = lds0 + (chunk & 1) * ASZ;
#pragma unroll
        for (int c = 0; c < NA; ++c) hdma16(asrc[c] ? asrc[c] + chunk * HK : zsrc, st + 16 * (wave_s * NA + c) * 64);
    };
    auto issue_a = [&](int t) {                           // this wave's rows of step t -> activation stage t % NSA
        const unsigned st = lds0 + (t % NSA) * ASZ;
        long off = (long)t * HK;
        int tap = 0;
        if constexpr (TAPS == 9) {
            tap = (int)__umulhi((unsigned)t, kmagic);
            if (s2d) {
                int rem, ftap, da, db;
                class_tap(t, rem, ftap, da, db);
                off = (long)(da * g.W + db) * g.lda + rem * HK;
            } else {
                const int a = tap / 3, b = tap - 3 * a;
                const int dh = g.flip ? 1 - a : a - 1, dw = g.flip ? 1 - b : b - 1;
                off = (long)(dh * g.Win + dw) * g.lda + (t - tap * kpt) * HK;
            }
        }
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const h16_t* src = asrc[c] + off;
            if constexpr (TAPS == 9) src = (tapmask[c] >> tap) & 1u ? src : zsrc;
#if PECLR_CONVH_NT & 1
            if (TAPS == 1 && nct == 1) hdma16_nt(src, st + (wave_s * RM + 16 * c) * 64);   // (no other workgroup reads these rows)
            else
#endif
            hdma16(src, st + (wave_s * RM + 16 * c) * 64);
        }
    };

    // order of issue: A(0), B(0), A(1) | step t: B(t + 1), A(t + 2).  The counter retires in order, so at the top of step t
    // "at most the NA row DMAs of A(t + 1) still in flight" means A(t) and B(t) have landed.
    PECLR_PHASE(0);
    if constexpr (RING != 0) {                            // RING: stage 0, then W chunks 0 .. NSB - 2 (step s = chunk * 9 + tap
        issue_ring(0);                                    // reads W chunk tap * kpt + chunk)
#pragma unroll
        for (int q = 0; q < NSB - 1; ++q) issue_b(q < nk ? (q % 9) * kpt + q / 9 : 0, q);
    } else {
        issue_a(0);
        issue_b(0, 0);
        if (nk > 1) issue_a(1);
    }
    // per-column constants of the epilogue, requested behind the first stages (their latency hides behind the main loop): the
    // statistics' shift of this lane's columns; the BatchNorm backward's mean / invstd / scale / shift of column tid (to LDS
    // after the loop).  Always NE load instructions, whatever the launch asks for, so that the DMA waits below stay exact
    // (without statistics they read the first floats of the packed weights)
    constexpr int NE = NTL + (BBF ? 4 : 0);
    float kshift[NTL];
    {
        const float* sp = g.stat_partial ? g.stat_shift + n0 : reinterpret_cast<const float*>(g.Bp);
#pragma unroll
        for (int y = 0; y < NTL; ++y) kshift[y] = sp[y * 32 + i];
    }
    float bcst[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (BBF) {
        const int c = n0 + (tid < PNL ? tid : PNL - 1);
        bcst[0] = g.bb_mean[c]; bcst[1] = g.bb_invstd[c]; bcst[2] = g.bb_ss[c]; bcst[3] = g.bb_ss[g.N + c];
    }
    // fragment (tile a, k-extent kk) of this lane: row a * 32 + i, chunk 2 kk + kh, at slot chunk ^ ((row >> 2) & 3)
    const int arow = wave * RM + i;
    const int asw = (i >> 2) & 3;
    if constexpr (RING != 0) {
        // The loop is written for its instruction count: a wave shares its SIMD with at most one other (64 KiB of LDS per
        // workgroup), and what it issues besides the 16 products of a step is not hidden (measured: 1 300 clocks per step with
        // the address arithmetic in the loop, 865 for the same requests / reads / products without it).  So the nine taps are
        // unrolled (their fragment offsets are nine registers computed once), the W requests take a scalar base + the lane's
        // 16 bytes, and every fragment read is base + immediate.
        // In-order counter: younger than W chunk s at the top of step s are the W chunks of steps s + 1 .. s + NSB - 2, the
        // next ring stage if it was requested after them (at tap 0, behind that step's W request: taps 1 .. NSB - 1), and in
        // step 0 the NE constants.  The last NSB - 2 steps simply wait for everything.
        constexpr int N0 = (NSB - 2) * NBD;
        unsigned aoff[9];                                  // byte offset of this lane's fragment (a = 0, kk = 0) in a stage, per tap
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ta = tap / 3, tb = tap - 3 * ta;
            const int dh = g.flip ? 1 - ta : ta - 1, dw = g.flip ? 1 - tb : tb - 1;
            const int r = arow + (Wp + 1) + dh * Wp + dw;  // (a = 1: + 32 rows = + 2048 bytes, same slot permutation; kk = 1: ^ 32)
            aoff[tap] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
        }
        // W chunk of (chunk, tap) = tap * kpt + chunk; this wave's pieces of it start at wbase + that * HCHUNK
        const unsigned char* wbase = static_cast<const unsigned char*>(g.Bp) + (size_t)(NTL == 4 ? ct : ct >> 1) * nk_all * HCHUNK +
                                     ((NTL == 4 ? 0 : (ct & 1) * 4) + (NTL == 4 ? 2 : 1) * wave_s) * 1024;
        const unsigned wlds = lds0 + B0 + (NTL == 4 ? 2 : 1) * wave_s * 1024;
        const unsigned lane16 = lane * 16;
        const unsigned tap_stride = (unsigned)kpt * HCHUNK;
        int bst = 0;                                       // W stage of the current step (= step % NSB)
        auto request_w = [&](int chunk, int tap, int stage) {
            const unsigned char* src = wbase + (size_t)((unsigned)tap * tap_stride) + (size_t)chunk * HCHUNK;
            const unsigned d = wlds + stage * BSZ;
            hdma16s<0>(src, lane16, d);
            if constexpr (NTL == 4) hdma16s<1024>(src, lane16, d);      // (the immediate moves BOTH addresses)
        };
        auto step = [&](auto tapc, int chunk) {
            constexpr int tap = decltype(tapc)::value;
            const bool last_chunk = chunk + 1 == kpt;
            if (tap + NSB - 1 > 9 && last_chunk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (tap == 0 && chunk == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0 + NE) : "memory");
            else if (tap >= 1 && tap <= NSB - 1 && !last_chunk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0 + NA) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N0) : "memory");
            __builtin_amdgcn_s_barrier();                 // publishes W chunk s (and the stage); every wave is done with step s - 1
            asm volatile("" ::: "memory");
            const unsigned char* sa = lds + (chunk & 1) * ASZ + aoff[tap];
            const unsigned char* sb = lds + B0 + bst * BSZ + lane16;
            uint4 af[2][WM], bf[2][NTL];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const unsigned char* sak = kk ? lds + (((unsigned)(sa - lds)) ^ 32u) : sa;
#pragma unroll
                for (int a = 0; a < WM; ++a) af[kk][a] = *reinterpret_cast<const uint4*>(sak + a * 2048);
#pragma unroll
                for (int y = 0; y < NTL; ++y) bf[kk][y] = *reinterpret_cast<const uint4*>(sb + (2 * y + kk) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int y = 0; y < NTL; ++y)
#pragma unroll
                    for (int a = 0; a < WM; ++a) acc[a][y] = H::mma(af[kk][a], bf[kk][y], acc[a][y]);
                if (kk == 0) {
                    // the next requests go out while the matrix cores work through the first half's products: W chunk
                    // s + NSB - 1 into the buffer step s - 1 read, and at tap 0 the next ring stage into the one chunk - 1 read
                    constexpr int t2 = (tap + NSB - 1) % 9, dc = (tap + NSB - 1) / 9;
                    int st2 = bst + NSB - 1;
                    st2 = st2 >= NSB ? st2 - NSB : st2;
                    if (chunk + dc < kpt) request_w(chunk + dc, t2, st2);
                    if (tap == 0 && !last_chunk) issue_ring(chunk + 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);            // (the products stay in their step)
            bst = bst + 1 == NSB ? 0 : bst + 1;
        };
        for (int chunk = 0; chunk < kpt; ++chunk) {
            step(std::integral_constant<int, 0>{}, chunk); step(std::integral_constant<int, 1>{}, chunk);
            step(std::integral_constant<int, 2>{}, chunk); step(std::integral_constant<int, 3>{}, chunk);
            step(std::integral_constant<int, 4>{}, chunk); step(std::integral_constant<int, 5>{}, chunk);
            step(std::integral_constant<int, 6>{}, chunk); step(std::integral_constant<int, 7>{}, chunk);
            step(std::integral_constant<int, 8>{}, chunk);
        }
    } else
    for (int t = 0; t < nk; ++t) {
        // A(t), B(t) have landed: younger than them are at most the NA row DMAs of A(t + 1) -- and, in step 0, the NE
        // constant loads issued behind the first stages
        if (t == 0) {
            if (nk > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NA + NE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NE) : "memory");
        } else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                     // publishes A(t), B(t); every wave is done reading step t - 1
        asm volatile("" ::: "memory");
        if (t + 1 < nk) issue_b(t + 1, t + 1);            // into the W buffer step t - 1 read
        if (t + 2 < nk) issue_a(t + 2);                   // into the row buffer step t - 1 read
        const unsigned char* sa = lds + (t % NSA) * ASZ;
        const unsigned char* sb = lds + B0 + (t % NSB) * BSZ + lane * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[WM];
#pragma unroll
            for (int a = 0; a < WM; ++a)
                af[a] = *reinterpret_cast<const uint4*>(sa + (arow + a * 32) * 64 + (((2 * kk + kh) ^ asw) << 4));
#pragma unroll
            for (int y = 0; y < NTL; ++y) {
                const uint4 bf = *reinterpret_cast<const uint4*>(sb + (2 * y + kk) * 1024);
#pragma unroll
                for (int a = 0; a < WM; ++a) acc[a][y] = H::mma(af[a], bf, acc[a][y]);
            }
        }
    }

    // ---- epilogue.  The stored value is the accumulator rounded to the 16-bit format; the fused BatchNorm sums are taken of
    // those rounded values (what a separate pass over the stored tensor would read).
    PECLR_PHASE(1);                                       // (0: set-up + first requests, 1: main loop, 2: statistics, 3: stores)
    __syncthreads();
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int y = 0; y < NTL; ++y)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                if (!ADD || !g.addend) {                  // (with an addend the rounding happens after the addition, below)
                    const unsigned p = H::pack2(acc[a][y][r], acc[a][y][r + 1]);
                    acc[a][y][r] = H::up(p & 0xFFFFu);
                    acc[a][y][r + 1] = H::up(p >> 16);
                }
            }
    float* sl = reinterpret_cast<float*>(lds + 4 * 32 * XE * 4);      // [wave][2][128] column sums
    if (g.stat_partial) {
        const bool full = !RING && row_block * TM + TM <= g.M;     // (uniform: whole tile inside the matrix)
        unsigned long long ring_ok = 0;                   // RING: bit r = row r of this wave's 64 is a pixel (not padding)
        if constexpr (RING != 0) ring_ok = __ballot(ring_row(m0 + lane) >= 0);
#pragma unroll
        for (int y = 0; y < NTL; ++y) {
            const float k0 = kshift[y];
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[a][y][r] - k0;
                    const bool in = RING ? (ring_ok >> (a * 32 + mfma32_row(r, kh))) & 1ull : m0 + a * 32 + mfma32_row(r, kh) < g.M;
                    if (full || in) { sum += d; sq = fmaf(d, d, sq); }
                }
            sum += __shfl_xor(sum, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (kh == 0) { sl[(wave * 2) * 128 + y * 32 + i] = sum; sl[(wave * 2 + 1) * 128 + y * 32 + i] = sq; }
        }
        __syncthreads();
        {
            const int which = tid >> 7, col = tid & 127;
            if (col < PNL) {
                const float v = ((sl[(0 * 2 + which) * 128 + col] + sl[(1 * 2 + which) * 128 + col]) +
                                 sl[(2 * 2 + which) * 128 + col]) + sl[(3 * 2 + which) * 128 + col];
                g.stat_partial[((size_t)row_block * 2 + which) * g.N + n0 + col] = v;
                if (row_block == 0 && which == 0)
                    g.stat_partial[(size_t)(((RING ? g.Mp : g.M) + TM - 1) / TM) * 2 * g.N + n0 + col] = g.stat_shift[n0 + col];

            }
        }
        __syncthreads();
    }
    PECLR_PHASE(2);
    float* cst = reinterpret_cast<float*>(lds + CST);                 // [mean | invstd | scale | shift][128]
    if (BBF && g.bb_partial) {
        if (tid < PNL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cst[q * 128 + tid] = bcst[q];
        }
        __syncthreads();
    }
    // wave-private 32 x 64 transposes: a lane then owns 8 consecutive columns of a row -> 16-byte loads and stores, 128
    // contiguous bytes per row
    float* wl = reinterpret_cast<float*>(lds + wave * (32 * XE * 4));
    const int er = lane >> 3, ec = (lane & 7) * 8;
    int om[WM][4];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = m0 + a * 32 + er + 8 * jj;
            om[a][jj] = m < g.M || RING ? m : -1;         // (-1: no such row)
            if constexpr (RING != 0) om[a][jj] = ring_row(m);
            if (s2d && m < g.M) {
                const int j2 = m % g.W, q = m / g.W, i2 = q % g.H, img = q / g.H;
                om[a][jj] = (img * 2 * g.H + 2 * i2 + ph) * (2 * g.W) + 2 * j2 + pw;
            }
        }
#pragma unroll
    for (int yp = 0; yp < NTL / 2; ++yp) {
        const int nt = n0 + yp * 64;
        float bmean[8], binv[8], bsc[8], bsh[8];
        float sb[8], sg[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) sb[q] = sg[q] = 0.f;
        if (BBF && g.bb_partial) {
#pragma unroll
            for (int q = 0; q < 8; q += 4) {
                const int cc = yp * 64 + ec + q;
                *reinterpret_cast<f32x4*>(bmean + q) = *reinterpret_cast<const f32x4*>(cst + cc);
                *reinterpret_cast<f32x4*>(binv + q) = *reinterpret_cast<const f32x4*>(cst + 128 + cc);
                *reinterpret_cast<f32x4*>(bsc + q) = *reinterpret_cast<const f32x4*>(cst + 256 + cc);
                *reinterpret_cast<f32x4*>(bsh + q) = *reinterpret_cast<const f32x4*>(cst + 384 + cc);
            }
        }
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            const int mt = m0 + a * 32;
            uint4 dv[4], xv[4];
            unsigned mb[4], ab[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = mt + er + 8 * jj;
                dv[jj] = make_uint4(0u, 0u, 0u, 0u);
                ab[jj] = 0u;
                if (ADD && g.addend && om[a][jj] >= 0) {
                    size_t arow = m;
                    bool has = true;
                    if (g.add_h) {
                        const int w = m % g.add_w, hq = m / g.add_w, h = hq % g.add_h, img = hq / g.add_h;
                        has = ((h | w) & 1) == 0;
                        arow = ((size_t)img * (g.add_h >> 1) + (h >> 1)) * (g.add_w >> 1) + (w >> 1);
                    }
                    if (has) {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4* asrc4 = reinterpret_cast<const u32x4*>(g.addend + arow * g.ldd + nt + ec);
                        const u32x4 tv = g.stream_out ? __builtin_nontemporal_load(asrc4) : *asrc4;
                        dv[jj] = make_uint4(tv[0], tv[1], tv[2], tv[3]);
                        ab[jj] = g.add_mask ? (g.add_mask[(size_t)m * (g.N >> 5) + ((nt + ec) >> 5)] >> (ec & 31)) & 0xFFu : 0xFFu;
                    }
                }
                if (BBF && g.bb_partial && om[a][jj] >= 0) {
                    xv[jj] = *reinterpret_cast<const uint4*>(g.bb_x + (size_t)om[a][jj] * g.N + nt + ec);
                    mb[jj] = g.bb_mask ? (g.bb_mask[(size_t)om[a][jj] * (g.N >> 5) + ((nt + ec) >> 5)] >> (ec & 31)) & 0xFFu : 0u;
                }
            }
#pragma unroll
            for (int y2 = 0; y2 < 2; ++y2)
#pragma unroll
                for (int r = 0; r < 16; ++r) wl[mfma32_row(r, kh) * XE + y2 * 32 + i] = acc[a][2 * yp + y2][r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                float c[8];
                *reinterpret_cast<float4*>(c) = *reinterpret_cast<const float4*>(wl + (er + 8 * jj) * XE + ec);
                *reinterpret_cast<float4*>(c + 4) = *reinterpret_cast<const float4*>(wl + (er + 8 * jj) * XE + ec + 4);
                // The epilogue of the fused entry gradients is bound by its vector instructions (four waves per SIMD each issuing
                // all of this per row): the launch's modes are decided per ROW, not per element, words are split with one
                // instruction per half, and the reduction's normalisation (x - mean) * invstd keeps its factor for the end.
                const unsigned dw[4] = {dv[jj].x, dv[jj].y, dv[jj].z, dv[jj].w};
                unsigned ow[4];
                if (ADD && g.addend) {
                    if (g.add_mask) {                     // 1-bit mask of the addend (an identity shortcut's (dy, mask) hand-over)
                        const unsigned bits = ab[jj];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            c[2 * q] += (bits >> (2 * q)) & 1u ? H::lo(dw[q]) : 0.f;
                            c[2 * q + 1] += (bits >> (2 * q + 1)) & 1u ? H::hi(dw[q]) : 0.f;
                        }
                    } else {                              // dense, or compact with zeros where the row has no entry
#pragma unroll
                        for (int q = 0; q < 4; ++q) { c[2 * q] += H::lo(dw[q]); c[2 * q + 1] += H::hi(dw[q]); }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) ow[q] = H::pack2(c[2 * q], c[2 * q + 1]);
                if (om[a][jj] >= 0) {
                    {
                        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                        u32x4* dst4 = reinterpret_cast<u32x4*>(g.out + (size_t)om[a][jj] * g.ldo + nt + ec);
                        const u32x4 tv = {ow[0], ow[1], ow[2], ow[3]};
                        if (g.stream_out) __builtin_nontemporal_store(tv, dst4);
                        else *dst4 = tv;
                    }
                    if (BBF && g.bb_partial) {
                        const unsigned xw[4] = {xv[jj].x, xv[jj].y, xv[jj].z, xv[jj].w};
                        float x[8], d[8];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            x[2 * q] = H::lo(xw[q]); x[2 * q + 1] = H::hi(xw[q]);
                            d[2 * q] = H::lo(ow[q]); d[2 * q + 1] = H::hi(ow[q]);       // (the rounded outputs)
                        }
                        if (g.bb_relu) {
                            if (g.bb_mask) {
                                const unsigned bits = mb[jj];
#pragma unroll
                                for (int q = 0; q < 8; ++q) d[q] = (bits >> q) & 1u ? d[q] : 0.f;
                            } else {
#pragma unroll
                                for (int q = 0; q < 8; ++q) d[q] = fmaf(x[q], bsc[q], bsh[q]) > 0.f ? d[q] : 0.f;
                            }
                        }
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            sb[q] += d[q];
                            sg[q] = fmaf(d[q], x[q] - bmean[q], sg[q]);
                        }
                    }
                }
            }
        }
        if (BBF && g.bb_partial) {                        // lanes with equal (lane & 7) hold the same eight columns
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                sg[q] *= binv[q];                         // (sum of d (x - mean), times invstd)
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { sb[q] += __shfl_xor(sb[q], o, 64); sg[q] += __shfl_xor(sg[q], o, 64); }
                if (er == 0) { sl[(wave * 2) * 128 + yp * 64 + ec + q] = sb[q]; sl[(wave * 2 + 1) * 128 + yp * 64 + ec + q] = sg[q]; }
            }
        }
    }
    if (BBF && g.bb_partial) {
        __syncthreads();
        const int which = tid >> 7, col = tid & 127;
        if (col < PNL) {
            const float v = ((sl[(0 * 2 + which) * 128 + col] + sl[(1 * 2 + which) * 128 + col]) +
                             sl[(2 * 2 + which) * 128 + col]) + sl[(3 * 2 + which) * 128 + col];
            const size_t rb = (size_t)row_block + (s2d ? (size_t)blockIdx.y * ((g.M + TM - 1) / TM) : 0);   // (RING: row blocks of the padded space)
            g.bb_partial[(rb * 2 + which) * g.N + n0 + col] = v;
        }
    }
    PECLR_PHASE(3);
}

// ---- weight packing: W[N][K] fp32 master weights (or their transpose) -> fragment-ordered 16-bit chunks.  One workgroup per
// (128 columns, 32 k) chunk, thread = (column, k-half): for each of the two 16-k extents 8 k-values -> 16 bytes.
struct HPackDesc {          // device table entry (8 x int64): as PackDesc of gemm_x6p.hip; pad = PECLR_DTYPE_BF16 / _F16
    int64_t src, dst, n, k, ld, transposed, chunk_begin, dtype;
};

__global__ __launch_bounds__(256) void h_pack_kernel(const HPackDesc* descs, int count) {
    static_assert(sizeof(HPackDesc) == 64 && offsetof(HPackDesc, chunk_begin) == 48, "pack_entry_of_chunk reads field 6 of 8 x int64 entries");
    const int d = pack_entry_of_chunk(reinterpret_cast<const int64_t*>(descs), count);
    const HPackDesc e = descs[d];
    const int chunk = (int)(blockIdx.x - e.chunk_begin);
    const int nks = (int)(e.k / HK);
    const int ct = chunk / nks, ks = chunk % nks;
    const int col = threadIdx.x & 127, kh = threadIdx.x >> 7;
    const int n = ct * HN + col;
    const float* src = reinterpret_cast<const float*>(e.src);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k0 = ks * HK + kk * 16 + 8 * kh;
        float v[8];
        if (n >= e.n) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = 0.f;
        } else if (e.transposed > 1) {
            const int taps = (int)e.transposed, cout = (int)(e.k / taps);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q, tap = k / cout, co = k - tap * cout;
                v[q] = src[((size_t)co * taps + tap) * e.ld + n];
            }
        } else if (e.transposed) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(size_t)(k0 + q) * e.ld + n];
        } else {
            const float4 lo = *reinterpret_cast<const float4*>(src + (size_t)n * e.ld + k0);
            const float4 hi = *reinterpret_cast<const float4*>(src + (size_t)n * e.ld + k0 + 4);
            v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
        }
        unsigned w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = e.dtype == PECLR_DTYPE_F16 ? F16::pack2(v[2 * q], v[2 * q + 1]) : BF16::pack2(v[2 * q], v[2 * q + 1]);
        unsigned char* dst = reinterpret_cast<unsigned char*>(e.dst) + (size_t)chunk * HCHUNK + ((col >> 5) * 2 + kk) * 1024 +
                             ((col & 31) + 32 * kh) * 16;
        *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

int pick_rows(int M, int N) {
    // These launches stream: what counts is workgroups per CU (each one's load -> multiply -> store phases are serial), and
    // 128-row tiles run four per CU (<= 128 VGPRs, 40 KiB of LDS) where 256-row tiles run two.  Probe at ResNet-50's shapes
    // (tools/exp/conv_h_probe.py): 128 rows win everywhere but one shape (K = 1024 -> N = 256: 40.6 vs 44.5 us).
    (void)M; (void)N;
    return 128;
}

void set_bb(HArgs& g, const peclr_bn_bwd_fuse* bb) {
    g.bb_x = bb ? reinterpret_cast<const h16_t*>(bb->x) : nullptr;     // (16-bit rows behind the struct's float pointer)
    g.bb_mean = bb ? bb->mean : nullptr;
    g.bb_invstd = bb ? bb->invstd : nullptr; g.bb_ss = bb ? bb->scale_shift : nullptr;
    g.bb_mask = bb ? bb->relu_mask : nullptr; g.bb_relu = bb ? bb->relu : 0; g.bb_partial = bb ? bb->partial : nullptr;
}

// rows of the ring stage a 3x3 / stride-1 launch over W-pixel rows needs (0: not eligible -- rows wider than 62 pixels, or
// switched off: PECLR_CONV3_RING=0 for A/B runs)
// (rows of 63 ... 126 pixels -- layer1 at 448 x 448 inputs, BASELINE config C5: 112 -- CAN take the ring form on 128-row tiles:
// 128 + 2 (W + 2) <= 384 rows per stage keep two stages + two W stages inside the 64 KiB an LDS-DMA can address.  Built and
// tested in round 5, and measured slower than the per-tap form there (C5 shape, bf16: 3x3 forward +29 us, input gradient +42 us
// per layer1 launch -- a stage fetches 2.8 rows per output row and only two workgroups fit a CU), so the library's own choice
// (tile_rows = 0) keeps the per-tap form for them; PECLR_CONV3_RING_WIDE=1 or an explicit PECLR_CONV_H_RING selects it.)
int ring_tile(int W) { return W <= 62 ? 256 : 128; }
int ring_rows(int taps, int stride, int s2d, int W) {
    static const int on = getenv("PECLR_CONV3_RING") ? atoi(getenv("PECLR_CONV3_RING")) : 1;
    if (!on || taps != 9 || stride != 1 || s2d || W > 126) return 0;
    if (W > 62) return 384;
    return 256 + 2 * (W + 2) <= 320 ? 320 : 384;
}
// ... and whether the library picks it when the caller leaves the choice (tile_rows = 0)
int ring_default(int taps, int stride, int s2d, int W) {
    static const int wide = getenv("PECLR_CONV3_RING_WIDE") ? atoi(getenv("PECLR_CONV3_RING_WIDE")) : 0;
    return (W <= 62 || wide) ? ring_rows(taps, stride, s2d, W) : 0;
}

// experiment switch (default off: the consumer of a 16-bit output -- the next BatchNorm pass -- finds part of it in the
// memory-side cache): PECLR_CONV_H_STREAM_OUT=<MB> writes outputs larger than that many MB (and reads their addends) non-temporally
static bool stream_past_caches_h(size_t bytes) {
    static const long mb = getenv("PECLR_CONV_H_STREAM_OUT") ? atol(getenv("PECLR_CONV_H_STREAM_OUT")) : 0;
    return mb > 0 && bytes > ((size_t)mb << 20);
}

template <typename H, int EP>
int launch_h(const HArgs& g, int tile_rows, int taps, hipStream_t stream) {
    if constexpr ((EP & 1) == 0) {
        if (tile_rows == PECLR_CONV_H_RING) {
            const int rr = ring_rows(taps, g.stride, g.s2d, g.W);
            if (!rr) return PECLR_ERR_UNSUPPORTED;
            const int tm = ring_tile(g.W);
            const int nrb = (g.Mp + tm - 1) / tm;
            const bool narrow = g.N % HN != 0;
            const dim3 grid(8 * ((nrb + 7) / 8) * (narrow ? g.N / 64 : g.N / HN));
            if (tm == 128) {
                if (narrow) hipLaunchKernelGGL((conv_h_kernel<H, 1, 9, 2, EP, 384>), grid, dim3(256), 0, stream, g);
                else hipLaunchKernelGGL((conv_h_kernel<H, 1, 9, 4, EP, 384>), grid, dim3(256), 0, stream, g);
            } else if (rr == 320) {
                if (narrow) hipLaunchKernelGGL((conv_h_kernel<H, 2, 9, 2, EP, 320>), grid, dim3(256), 0, stream, g);
                else hipLaunchKernelGGL((conv_h_kernel<H, 2, 9, 4, EP, 320>), grid, dim3(256), 0, stream, g);
            } else {
                if (narrow) hipLaunchKernelGGL((conv_h_kernel<H, 2, 9, 2, EP, 384>), grid, dim3(256), 0, stream, g);
                else hipLaunchKernelGGL((conv_h_kernel<H, 2, 9, 4, EP, 384>), grid, dim3(256), 0, stream, g);
            }
            return launch_status();
        }
    }
    if (tile_rows == PECLR_CONV_H_RING) return PECLR_ERR_UNSUPPORTED;
    const int nrb = (g.M + tile_rows - 1) / tile_rows;
    static const int force_narrow = getenv("PECLR_CONV_H_NARROW") ? atoi(getenv("PECLR_CONV_H_NARROW")) : 0;   // experiments
    // the fused entry gradient (addend + BatchNorm reduction: three streams in the epilogue) runs best on 64-column tiles at
    // four workgroups per CU (probe: 392 -> 315, 218 -> 178, 111 -> 93, 77 -> 68 us at layers 1 - 4)
    const bool narrow = g.N % HN != 0 || force_narrow == 1 || (EP == 3 && force_narrow != 2);
    const dim3 grid(8 * ((nrb + 7) / 8) * (narrow ? g.N / 64 : g.N / HN), taps == 9 && g.s2d ? 4 : 1);
#define PECLR_LAUNCH(WM_, TAPS_)                                                                                    \
    do {                                                                                                            \
        if (narrow) hipLaunchKernelGGL((conv_h_kernel<H, WM_, TAPS_, 2, EP>), grid, dim3(256), 0, stream, g);       \
        else hipLaunchKernelGGL((conv_h_kernel<H, WM_, TAPS_, 4, EP>), grid, dim3(256), 0, stream, g);              \
    } while (0)
    if constexpr ((EP & 1) == 0) {                        // (an addend only exists for the 1x1 products)
        if (taps == 9) { if (tile_rows == 256) PECLR_LAUNCH(2, 9); else PECLR_LAUNCH(1, 9); return launch_status(); }
    }
    if (taps == 9) return PECLR_ERR_UNSUPPORTED;
    if (tile_rows == 256) PECLR_LAUNCH(2, 1); else PECLR_LAUNCH(1, 1);
#undef PECLR_LAUNCH
    return launch_status();
}

template <typename H>
int launch_ep(const HArgs& g, int tile_rows, int taps, hipStream_t stream) {
    const int ep = (g.addend ? 1 : 0) | (g.bb_partial ? 2 : 0);
    switch (ep) {
        case 0: return launch_h<H, 0>(g, tile_rows, taps, stream);
        case 1: return launch_h<H, 1>(g, tile_rows, taps, stream);
        case 2: return launch_h<H, 2>(g, tile_rows, taps, stream);
        default: return launch_h<H, 3>(g, tile_rows, taps, stream);
    }
}

int dispatch(int dtype, const HArgs& g, int tile_rows, int taps, hipStream_t stream) {
    if (dtype == PECLR_DTYPE_BF16) return launch_ep<BF16>(g, tile_rows, taps, stream);
    if (dtype == PECLR_DTYPE_F16) return launch_ep<F16>(g, tile_rows, taps, stream);
    return PECLR_ERR_UNSUPPORTED;
}

}  // namespace
}  // namespace peclr

using namespace peclr;

#ifdef PECLR_CONV_H_TIMING
extern "C" int peclr_debug_conv_h_timing(unsigned long long* out, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(conv_h_timing), sizeof(conv_h_timing));
    if (reset) { unsigned long long z[8] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(conv_h_timing), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -1;
}
#endif

extern "C" int64_t peclr_h_pack_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % 64 || K % HK) return 0;
    return (int64_t)((N + HN - 1) / HN * HN) * K * 2;
}

extern "C" int peclr_h_pack(const void* desc_table, int count, int total_chunks, peclr_stream_t stream) {
    if (!desc_table) return PECLR_ERR_NULL;
    if (count <= 0 || total_chunks <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(h_pack_kernel, dim3(total_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const HPackDesc*>(desc_table), count);
    return launch_status();
}

extern "C" int peclr_conv_h_tile_rows(int M, int N) {
    if (M <= 0 || N <= 0 || N % 64) return 0;
    return pick_rows(M, N);
}

// Rows of the partial-sum tables (BatchNorm statistics / backward reduction) a peclr_conv_h launch with this tile_rows argument
// writes: row blocks of the output pixels, or -- ring launches (tile_rows 0 picks them where they apply, PECLR_CONV_H_RING asks
// for them) -- 256-pixel (rows of 63 ... 126 pixels: 128-pixel) blocks of the padded space NB x (H + 1) x (W + 1).  0: unsupported.
extern "C" int peclr_conv_h_row_blocks(int NB, int H, int W, int Cout, int taps, int stride, int tile_rows) {
    if (NB <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (stride != 1 && stride != 2) || H % stride || W % stride) return 0;
    const int Ho = H / stride, Wo = W / stride;
    const long M = (long)NB * Ho * Wo;
    const int rr = ring_rows(taps, stride, 0, Wo);
    if (tile_rows == 0) tile_rows = ring_default(taps, stride, 0, Wo) ? PECLR_CONV_H_RING : pick_rows((int)M, Cout);
    if (tile_rows == PECLR_CONV_H_RING) return rr ? (int)(((long)NB * (Ho + 1) * (Wo + 1) + ring_tile(Wo) - 1) / ring_tile(Wo)) : 0;
    if (tile_rows != 128 && tile_rows != 256) return 0;
    return (int)((M + tile_rows - 1) / tile_rows);
}

// 1x1 / stride-1 product with the optional epilogues: dense addend (add_h = 0), compact stride-2 addend (add_h, add_w > 0),
// 1-bit mask on the dense addend, BatchNorm statistics of the output, BatchNorm backward reduction.
extern "C" int peclr_gemm_h(int dtype, int M, int N, int K, const void* A, int lda, const void* Bp, void* C, int ldc,
                            const void* addend, int ldd, int add_h, int add_w, const unsigned* addend_mask, int tile_rows,
                            const float* stat_shift, float* stat_partial, const peclr_bn_bwd_fuse* bb, peclr_stream_t stream) {
    if (!A || !Bp || !C || (stat_partial && !stat_shift)) return PECLR_ERR_NULL;
    if (bb && (stat_partial || !bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial || ldc != N)) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % HK) return PECLR_ERR_SHAPE;
    if (add_h && (!addend || add_h < 2 || add_w < 2 || add_h % 2 || add_w % 2 || M % (add_h * add_w))) return PECLR_ERR_SHAPE;
    if (addend_mask && (!addend || add_h || N % 32)) return PECLR_ERR_SHAPE;
    if (lda % 8 || lda < K || ldc % 8 || ldc < N || (addend && (ldd % 8 || ldd < N))) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(Bp) || !aligned16(C) || (addend && !aligned16(addend))) return PECLR_ERR_ALIGN;
    if (tile_rows == 0) tile_rows = pick_rows(M, N);
    if (tile_rows != 128 && tile_rows != 256) return PECLR_ERR_UNSUPPORTED;
    HArgs g;
    g.A = static_cast<const h16_t*>(A); g.Bp = Bp; g.addend = static_cast<const h16_t*>(addend); g.out = static_cast<h16_t*>(C);
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldo = ldc; g.ldd = ldd; g.add_h = add_h; g.add_w = add_w; g.add_mask = addend_mask;
    g.stat_shift = stat_shift; g.stat_partial = stat_partial;
    g.H = g.W = 1; g.flip = 0; g.zeros = nullptr; g.stride = 1; g.Hin = g.Win = 1; g.s2d = 0; g.Mp = 0;
    set_bb(g, bb);
    g.stream_out = stream_past_caches_h((size_t)M * N * 2);
    return dispatch(dtype, g, tile_rows, 1, static_cast<hipStream_t>(stream));
}

// 3x3 / padding 1 (taps = 9) or 1x1 (taps = 1) convolution of an NHWC tensor, stride 1 or 2: forward (flip = 0) or, stride 1,
// the input gradient (flip = 1, X = dY, planes packed with transposed = 9).
extern "C" int peclr_conv_h(int dtype, int NB, int H, int W, int Cin, int Cout, int taps, int stride, const void* X, const void* Bp,
                            void* Y, int flip, int tile_rows, const void* zeros, const float* stat_shift, float* stat_partial,
                            const peclr_bn_bwd_fuse* bb, peclr_stream_t stream) {
    if (!X || !Bp || !Y || !zeros || (stat_partial && !stat_shift)) return PECLR_ERR_NULL;
    if (bb && (stat_partial || !bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial || Cout % 32)) return PECLR_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cout % 64 || Cin % HK || (taps != 1 && taps != 9) ||
        (stride != 1 && stride != 2) || (stride == 2 && (H % 2 || W % 2 || flip || bb)) || (taps == 1 && stride == 1))
        return PECLR_ERR_SHAPE;
    if ((long)NB * H * W > 0x7FFFFFFFL / 2) return PECLR_ERR_SHAPE;
    if (!aligned16(X) || !aligned16(Bp) || !aligned16(Y) || !aligned16(zeros)) return PECLR_ERR_ALIGN;
    const int Ho = H / stride, Wo = W / stride, M = NB * Ho * Wo;
    if (tile_rows == 0) tile_rows = ring_default(taps, stride, 0, Wo) ? PECLR_CONV_H_RING : pick_rows(M, Cout);
    if (tile_rows != 128 && tile_rows != 256 && tile_rows != PECLR_CONV_H_RING) return PECLR_ERR_UNSUPPORTED;
    if (tile_rows == PECLR_CONV_H_RING && ((long)NB * (H + 1) * (W + 1) > 0x7FFFFFFFL / 2 || !ring_rows(taps, stride, 0, Wo)))
        return PECLR_ERR_UNSUPPORTED;
    HArgs g;
    g.Mp = NB * (Ho + 1) * (Wo + 1);
    g.A = static_cast<const h16_t*>(X); g.Bp = Bp; g.addend = nullptr; g.out = static_cast<h16_t*>(Y);
    g.M = M; g.N = Cout; g.K = taps * Cin; g.lda = Cin; g.ldo = Cout; g.ldd = Cout; g.add_h = g.add_w = 0; g.add_mask = nullptr;
    g.stat_shift = stat_shift; g.stat_partial = stat_partial;
    g.H = Ho; g.W = Wo; g.flip = flip ? 1 : 0; g.zeros = static_cast<const h16_t*>(zeros); g.stride = stride; g.Hin = H; g.Win = W; g.s2d = 0;
    set_bb(g, bb);
    g.stream_out = stream_past_caches_h((size_t)M * Cout * 2);
    return dispatch(dtype, g, tile_rows, taps, static_cast<hipStream_t>(stream));
}

// Input gradient of the 3x3 / padding-1 / stride-2 convolution: four dense implicit GEMMs, one per parity class of input pixels.
extern "C" int peclr_conv3x3_s2_dgrad_h(int dtype, int NB, int Ho, int Wo, int Cout, int Cin, const void* dY, const void* Bp, void* dX,
                                        int tile_rows, const void* zeros, const peclr_bn_bwd_fuse* bb, peclr_stream_t stream) {
    if (!dY || !Bp || !dX || !zeros) return PECLR_ERR_NULL;
    if (bb && (!bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial)) return PECLR_ERR_NULL;
    if (NB <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % HK) return PECLR_ERR_SHAPE;
    if ((long)NB * Ho * Wo * 4 > 0x7fffffffL) return PECLR_ERR_SHAPE;
    if (!aligned16(dY) || !aligned16(Bp) || !aligned16(dX) || !aligned16(zeros)) return PECLR_ERR_ALIGN;
    const int M = NB * Ho * Wo;
    if (tile_rows == 0) tile_rows = pick_rows(M, Cin);
    if (tile_rows != 128 && tile_rows != 256) return PECLR_ERR_UNSUPPORTED;
    HArgs g;
    g.A = static_cast<const h16_t*>(dY); g.Bp = Bp; g.addend = nullptr; g.out = static_cast<h16_t*>(dX);
    g.M = M; g.N = Cin; g.K = 9 * Cout; g.lda = Cout; g.ldo = Cin; g.ldd = Cin; g.add_h = g.add_w = 0; g.add_mask = nullptr;
    g.stat_shift = nullptr; g.stat_partial = nullptr;
    g.H = Ho; g.W = Wo; g.flip = 1; g.zeros = static_cast<const h16_t*>(zeros); g.stride = 1; g.Hin = Ho; g.Win = Wo; g.s2d = 1; g.Mp = 0;
    set_bb(g, bb);
    g.stream_out = stream_past_caches_h((size_t)M * 4 * Cin * 2);
    return dispatch(dtype, g, tile_rows, 9, static_cast<hipStream_t>(stream));
}
