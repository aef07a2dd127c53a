// fp32 GEMM on the bf16 matrix cores, second generation: C[M,N] = A[M,K] . W[N,K]^T (+ addend), fp32 in / fp32 out,
// with the WEIGHT operand split once per optimiser step instead of once per workgroup.
//
// Same arithmetic as gemm_x6.hip (x == h + m + l exactly, three bf16 numbers, common.hpp split3_pk; six of the nine partial
// products on v_mfma_f32_32x32x16_bf16 with fp32 accumulation, smallest first), different data path:
//
//   * W is a parameter: it changes once per step and is used by thousands of workgroups (forward, input gradient,
//     the fused entry gradient).  peclr_x6_pack_f32 splits it ONCE into three bf16 planes stored in MFMA fragment
//     order -- per (128 output columns, 16 k) one contiguous 12 KiB chunk of twelve 1 KiB pieces [32-column block]
//     [plane], a piece being lane l's 16 bytes at l * 16: columns n0 + (l & 31), k0 + 8 * (l >> 5) ... + 7.
//     The GEMM brings a chunk into LDS with twelve `global_load_lds_dwordx4` (LDS-DMA: no VGPRs, no VALU, no
//     ds_write, lane-linear = fragment order, bank-conflict free) and reads B fragments with ds_read_b128.
//   * The ACTIVATION operand still has to be split in the kernel (its producer is an HBM-bound BatchNorm pass that
//     cannot afford 6 more bytes per element).  Each wave owns 32 * WM rows x all 128 columns of the workgroup tile
//     (WM x 4 MFMA tiles, six products each), so a row is split ONCE per 128 output columns by exactly one wave and
//     its planes never leave that wave: fp32 rows arrive by LDS-DMA in a wave-private raw buffer, the wave reads them
//     back (same lane that the DMA wrote), splits in registers and writes three [k-half][row][8 k] planes into a
//     wave-private LDS region -- no workgroup barrier for A at all, in-order LDS execution of one wave orders the
//     fragment reads of k-step t before the plane writes of k-step t + 1 (single buffer).
//   * One raw s_barrier per k-step (16 k, 24 * WM MFMAs per wave) publishes the double-buffered B chunk; DMA waits are
//     counted (`s_waitcnt vmcnt(n)`), never drained in the loop.
//   LDS: 24 KiB (B, two chunks) + 4 waves x (3 planes + raw buffer) = 64.8 KiB at WM = 2 -> two workgroups per CU.
//
// The store path that bounded gemm_x6_nt128_kernel (six plane stores per operand element pair through the
// VGPR -> LDS port: 49 KiB per 1536 MFMA cycles) carries 24 KiB per 3072 MFMA cycles here.
#include <type_traits>

#include "common.hpp"

#include <cstddef>

namespace peclr {
namespace {

typedef uint16_t bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// A/B builds: bit 0 = the A rows of 1x1 products with ONE column tile, bit 1 = the epilogue's BatchNorm-x rows load with the non-temporal
// hint.  Measured neutral (round 5, same box: every tag within 1 us; on ALL A rows, shared ones included: + 1.5 ms per step): these
// kernels are not bound by their read path -- unlike the streaming BatchNorm passes (bn2d.hip), where the hint is worth 8 - 23 %
#ifndef PECLR_X6P_NT
#define PECLR_X6P_NT 0
#endif
constexpr int PN = 128;                  // output columns per workgroup (and per packed chunk)
constexpr int PK = 16;                   // k per step = one MFMA k-extent
constexpr int CHUNK3 = 12 * 1024;        // bytes of packed B per (128 columns, 16 k): three bf16 planes

struct X6PArgs {
    const float* A;
    const void* Bp;                      // packed planes of W (peclr_x6_pack_f32)
    const float* addend;
    float* out;
    int M, N, K, lda, ldo, ldd;
    // add_h > 0: the rows are the pixels of add_h x add_w images and the addend holds every second pixel only
    // ([images][add_h / 2][add_w / 2][ldd]: the input gradient of a 1x1 / stride-2 convolution, kept compact): rows at even
    // (h, w) add addend[(h / 2, w / 2)], the others nothing
    int add_h, add_w;
    // optional 1-bit mask of the addend ([M][N / 32] words, bit c % 32 of word c / 32): addend elements whose bit is clear
    // count as zero -- the gradient behind a ReLU, rectified here instead of in a pass of its own
    const unsigned* add_mask;
    int stream_out;
    // optional BatchNorm statistics of the OUTPUT (the convolution's BatchNorm2d in training mode): per workgroup row
    // block, per column, sum and sum of squares of (C - shift) over the block's rows -> stat_partial[row block][2][N],
    // plus the shift itself in row [n row blocks] -- the layout peclr_bn2d_finalize_f32 combines
    const float* stat_shift;
    float* stat_partial;
    // TAPS = 9 (3x3 convolution, stride 1, padding 1, as an implicit GEMM over NHWC rows): A is the activation [M = N*H*W][lda
    // = Cin], K = 9 * Cin ordered (tap, channel), tap = 3 * a + b reads the pixel at (+a-1, +b-1) -- (1-a, 1-b) when
    // `flip` (the input gradient's correlation with the flipped filter) -- or zeros outside the image
    int H, W, flip;
    // stride = 2 (forward only): the rows are the pixels of an H x W OUTPUT image, the activation has 2H' x 2W' = Hin x Win
    // pixels; output pixel (oh, ow) reads input pixel (2 oh + dh, 2 ow + dw) -- the strided 3x3 of a layer's first block
    // (TAPS = 9) and its 1x1 downsample convolution (TAPS = 1, dh = dw = 0)
    int stride, Hin, Win;
    // s2d = 1 (TAPS = 9): the INPUT gradient of the 3x3 / padding-1 / stride-2 convolution, one parity class of input pixels
    // per blockIdx.y.  A = dY [images * H * W][lda = Cout] over the H x W OUTPUT grid, out = dX over 2H x 2W.  Input pixel
    // (2 i + ph, 2 j + pw) receives filter row a only where ph + 1 - a is even: a = 1 from dY row i (ph = 0); a = 0 from
    // row i + 1 and a = 2 from row i (ph = 1) -- likewise for columns: 1, 2, 2 or 4 taps per class instead of 9 with
    // three quarters of them zero.  The M rows of a class are the (image, i, j) of the output grid; planes as for the
    // stride-1 input gradient (K order (tap, Cout)).
    int s2d;
    const float* zeros;                  // >= 64 bytes of zeros (source of the padding pixels)
    // NP = 2 instantiations ("pair" arithmetic, common.hpp split2_pk): Bp holds TWO fp16 planes per chunk (peclr_x6_pack_pair_f32),
    // *w_scale the power of two they were multiplied by; *a_absmax = max |A| over the WHOLE activation tensor (written by the pass
    // that produced it), from which every workgroup derives the same power of two for A
    const float* a_absmax;
    const float* w_scale;
    // optional: C is the gradient dY arriving at a BatchNorm2d(+ReLU) layer (this GEMM is the input gradient of the
    // convolution that consumed that layer's output).  The epilogue then performs the layer's backward REDUCTION on the tile
    // it holds: per row block and column, sum of dY' and of dY' * xhat with dY' = dY where the ReLU passed (recomputed from
    // the layer's input x, or read from its 1-bit mask), xhat = (x - mean) * invstd -> bb_partial[row block][2][N], the
    // layout peclr_bn2d_bwd_finalize_f32 combines; peclr_bn2d_bwd_reduce (a pass over dY and x) is not needed
    const float* bb_x;                   // the BatchNorm layer's input [M][N] (ld = N)
    const float* bb_mean;
    const float* bb_invstd;
    const float* bb_ss;                  // [2][N] scale, shift of the forward
    const unsigned* bb_mask;             // [M][N / 32] ReLU bit mask (layers with a residual added before the ReLU), or null
    int bb_relu;
    float* bb_partial;
};

__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int NP>
__device__ __forceinline__ f32x16 mman(const uint4& a, const uint4& b, f32x16 acc) {
    if constexpr (NP == 3) return mma(a, b, acc);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
// 16 bytes per lane, global -> LDS at (wave-uniform) dst + lane * 16.  Issued through inline assembly on purpose: for
// the builtin, hipcc's wait-count pass makes EVERY later ds_read wait for the DMA (vmcnt(0) right behind the issue --
// LDS accesses carry no alias information that would tell the B buffer being filled from the one being read), which
// serialises the pipeline.  Here the compiler does not know the instruction touches the vm counter; every wait on
// it is written by hand below (and no other VMEM instruction is in flight while DMAs are).
__device__ __forceinline__ void dma16(const void* src, unsigned lds_byte_offset) {   // offset: wave-uniform, in an SGPR
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds_byte_offset) : "memory", "m0");
}
#define PECLR_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// WM: 32-row MFMA tiles per wave (2 -> 256 x 128 workgroup tile, 1 -> 128 x 128 for problems with few row blocks)
// AREG: the fp32 rows travel global -> registers (inline-asm loads, hand-counted) instead of global -> LDS (DMA) -> registers
// ABL: ablation switches for tools/exp/x6p_ablate.hip (0 in the library): 1 no split / plane stores in the loop, 2 no raw-row
// loads either, 4 no B DMA in the loop, 8 no MFMAs, 16 no barrier, 32 B fragments always from buffer 0 (bits combine)
// NTL: 32-column MFMA tiles per wave (4 -> 128 output columns per workgroup; 2 -> 64, for 64-channel layers: half a packed chunk)
// HALO (TAPS = 9, stride 1): per 16-channel chunk the workgroup loads and splits the TM + 2 W + 2 pixels its nine taps touch
// ONCE into shared planes [plane][k-half][pixel][8 k] and every tap reads its A fragments from there at the tap's pixel offset
// (border rows from a 16-byte zero slot) -- instead of each wave loading and splitting its rows once per tap (nine times).
// K order of the loop: (chunk, tap); the packed filter chunks are indexed tap * chunks + chunk as before.  W <= 62.
// (Two more forms were built in round 5, measured slower and taken out of the library in round 6 -- persistent workgroups walking the
// tile index space, and the BatchNorm + ReLU in front of A applied in the row split: tools/exp/x6p_persist_tra.patch, numbers in
// docs/history.md E.)
// NP: planes per operand.  3 = the exact bf16 triple, six products (PECLR_X6).  2 = the fp16 pair of the scaled operands, three
// products (PECLR_X2): two thirds of the plane traffic through the LDS, half the matrix-core work, a cheaper split.
template <int WM, int ABL = 0, bool AREG = true, bool ILV = true, int TAPS = 1, int NTL = 4, bool HALO = false, int NP = 3>
__global__ __launch_bounds__(256, 2) void gemm_x6p_kernel(X6PArgs g) {
    static_assert(NP == 3 || (NP == 2 && AREG), "planes per operand: bf16 triple or fp16 pair");
    constexpr int CHUNK = NP * 4 * 1024;                 // bytes of packed B per (128 columns, 16 k): [32-column block][plane] pieces of 1 KiB
    constexpr int RM = 32 * WM;                          // rows per wave
    constexpr int TM = 4 * RM;                           // rows per workgroup
    constexpr int NRAW = RM / 16;                        // 1 KiB pieces of fp32 rows per wave and k-step
    constexpr int NB = 3;                                // B chunks in LDS (two k-steps of DMA run-ahead)
    constexpr int HALF = RM * 16 + 64;                   // bytes of one k-half of a plane (+64: the two halves of a row
                                                         // land on different banks for the 8-byte plane stores)
    constexpr int PLANE = 2 * HALF;
    constexpr int XEPL = 36;                             // floats per row of the epilogue's 32 x 32 transpose buffer
    constexpr int WAVE_PL = NP * PLANE;
    // bytes of one filter buffer: a 64-column workgroup lands only its six 1 KiB pieces of a chunk (18 instead of 36 KiB of
    // buffers: with its 120 VGPRs a fourth workgroup per CU; the halo variant's 128-row tile: a third)
    constexpr int CHL = CHUNK * NTL / 4;
    constexpr int RAW0 = NB * CHL < 5 * 32 * 36 * 4 ? 5 * 32 * 36 * 4 : NB * CHL;   // (the epilogue's transposes + sums live here: 22.5 KiB)
    constexpr int PL0 = RAW0 + (AREG ? 0 : 4 * RM * 64);   // DMA targets first (LDS-DMA addresses < 64 KiB)
    constexpr int NPXM = TM + 2 * 62 + 2;                // HALO: pixels of the patch at most (W <= 62: six 16-byte loads per thread at TM = 256)
    // k-half pitch: the pixels' 16 bytes + padding to 64 bytes past a multiple of 128 -- ds_write_b64 is served in groups of 16
    // lanes on 32 banks of 4 bytes, and a group of the patch store holds 4 pixels x both k-halves: with the bare pitch
    // (1528 dwords = 24 mod 32 at TM = 256) the second half's banks overlapped the first's, a 2-way conflict on every store
    // (0.187 of the kernel's LDS cycles, profiles/r04f_fp32_bench_mfma.txt; every other six-product kernel: <= 0.06)
    constexpr int PHALF = (NPXM * 16 + 63) / 128 * 128 + 64, PPLANE = 2 * PHALF, PATCH = NP * PPLANE;
    constexpr int ZOFF = RAW0 + PATCH;                   // HALO: 16 bytes of zeros
    // + 2.5 KiB at the end: the epilogue's per-column constants ([shift | mean | invstd | scale | shift'][128] floats), fetched
    // BEFORE the main loop -- in the epilogue each of the four column tiles used to wait a full memory round trip for them
    constexpr int CST0 = HALO ? ZOFF + 64 : PL0 + 4 * WAVE_PL;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[CST0 + 5 * 128 * 4];
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    constexpr int PNL = 32 * NTL;                        // output columns per workgroup
    const int nct = (g.N + PNL - 1) / PNL;
    f32x4 ar[NRAW];                                       // AREG: the next k-step's rows, in flight / landed
    const int vb = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, kh = lane >> 5;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lptr_t)lds;        // LDS byte address of the array
    unsigned char* const planes = lds + PL0 + wave * WAVE_PL;
    unsigned char* const raw = lds + RAW0 + wave * (RM * 64);
    const unsigned raw_a = __builtin_amdgcn_readfirstlane(lds0 + RAW0 + wave * (RM * 64));
    const unsigned b_a = __builtin_amdgcn_readfirstlane(lds0);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);             // wave-uniform copy in an SGPR (LDS-DMA targets)

    const int j = vb / 8;
    const int row_block = 8 * (j / nct) + vb % 8;         // all column tiles of a row block on one XCD
    if (row_block * TM >= g.M) return;
    const int m0 = row_block * TM + wave * RM, ct = j % nct, n0 = ct * PNL;
    // s2d: parity class of this workgroup (the four-tap class first: longest workgroups first)
    const bool s2d = TAPS == 9 && g.s2d;
    const int ph = s2d ? 1 - (int)(blockIdx.y >> 1) : 0, pw = s2d ? 1 - (int)(blockIdx.y & 1) : 0;
    const int ntap = s2d ? (1 + ph) * (1 + pw) : TAPS;
    const int nk = s2d ? ntap * (g.lda / PK) : g.K / PK;
    const int nk_all = g.K / PK;                                      // k-steps of a column tile's packed chunks
    // packed chunk of this column tile (NTL = 2: the first or second half of a 128-column chunk's twelve pieces)
    const int piece0 = NTL == 4 ? 0 : (ct & 1) * 2 * NP;
    const unsigned char* bsrc = static_cast<const unsigned char*>(g.Bp) + (size_t)(NTL == 4 ? ct : ct >> 1) * nk_all * CHUNK +
                                piece0 * 1024 + lane * 16;

    f32x16 acc[WM][NTL];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int y = 0; y < NTL; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][y][r] = 0.f;

    float pair_s = 1.f, pair_inv = 1.f;                   // NP = 2: the activation's power of two, and what undoes both operands' scaling
    if constexpr (NP == 2) {
        pair_s = pair_scale(*g.a_absmax);
        pair_inv = 1.f / (pair_s * *g.w_scale);           // (exact: powers of two, exponents within +-50 each)
    }
    float* const cst = reinterpret_cast<float*>(lds + CST0);
    {
        constexpr int PNL0 = 32 * NTL;
        const int c = (int)(blockIdx.x / 8 % ((g.N + PNL0 - 1) / PNL0)) * PNL0 + (tid < PNL0 ? tid : PNL0 - 1);
        float k0 = 0.f, k1 = 0.f, k2 = 0.f, k3 = 0.f, k4 = 0.f;
        if (g.stat_partial) k0 = g.stat_shift[c];
        if (g.bb_partial) { k1 = g.bb_mean[c]; k2 = g.bb_invstd[c]; k3 = g.bb_ss[c]; k4 = g.bb_ss[g.N + c]; }
        if (tid < PNL0) { cst[tid] = k0; cst[128 + tid] = k1; cst[256 + tid] = k2; cst[384 + tid] = k3; cst[512 + tid] = k4; }
    }
    if constexpr (!HALO) {
    // this lane's fp32 source: rows (lane >> 2) + 16 c of the wave's block, k-quad lane & 3 (rows past M re-read row M - 1)
    const float* asrc[NRAW];
    unsigned tapmask[NRAW];                               // TAPS = 9: bit tap = that tap's pixel lies inside the image
#pragma unroll
    for (int c = 0; c < NRAW; ++c) {
        int row = m0 + 16 * c + (lane >> 2);
        row = row < g.M ? row : g.M - 1;
        asrc[c] = g.A + (size_t)row * g.lda + 4 * (lane & 3);
        tapmask[c] = 0;
        if (TAPS == 9 || g.stride == 2) {
            const int ow = row % g.W, oh = (row / g.W) % g.H, img = row / (g.W * g.H);
            const int ih = g.stride * oh, iw = g.stride * ow;                 // centre pixel in the input image
            if (g.stride == 2) asrc[c] = g.A + ((size_t)(img * g.Hin + ih) * g.Win + iw) * g.lda + 4 * (lane & 3);
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
    const float* const zsrc = TAPS == 9 ? g.zeros + 4 * (lane & 3) : nullptr;
    const int kpt = TAPS == 9 ? g.lda / PK : 1;            // k-steps per tap
    // s2d: k-step t of the class = step `rem` of class tap u = filter tap (a, b), read from dY at (+da, +db)
    auto class_tap = [&](int t, int& rem, int& ftap, int& da, int& db) {
        const int u = t / kpt;
        rem = t - u * kpt;
        const int ua = u / (1 + pw), ub = u - ua * (1 + pw);
        da = ph && ua == 0;
        db = pw && ub == 0;
        ftap = 3 * (ph ? 2 * ua : 1) + (pw ? 2 * ub : 1);
    };
    auto issue_b = [&](int t) {                           // this wave's pieces (of 3 NTL) of chunk t -> buffer t % NB
        int bt_step = t;
        if constexpr (TAPS == 9) {
            if (s2d) {
                int rem, ftap, da, db;
                class_tap(t, rem, ftap, da, db);
                bt_step = ftap * kpt + rem;
            }
        }
        const unsigned char* s = bsrc + (size_t)bt_step * CHUNK;
        const unsigned d = b_a + (t % NB) * CHL;
        if constexpr (NTL == 4) {
#pragma unroll
            for (int q = 0; q < NP; ++q) dma16(s + (NP * wave_s + q) * 1024, d + (NP * wave_s + q) * 1024);
        } else {                                          // 2 NP pieces: (six: waves 0, 1 two each, waves 2, 3 one; four: one each)
            dma16(s + wave_s * 1024, d + wave_s * 1024);
            if (NP == 3 && wave_s < 2) dma16(s + (4 + wave_s) * 1024, d + (4 + wave_s) * 1024);
        }
    };
    auto issue_a = [&](int t) {
        long off = (long)t * PK;                          // floats from the row's first channel
        int tap = 0;
        if constexpr (TAPS == 9) {
            if (s2d) {
                int rem, ftap, da, db;
                class_tap(t, rem, ftap, da, db);
                tap = t / kpt;
                off = (long)(da * g.W + db) * g.lda + rem * PK;
            } else {
                tap = t / kpt;
                const int a = tap / 3, b = tap - 3 * a;
                const int dh = g.flip ? 1 - a : a - 1, dw = g.flip ? 1 - b : b - 1;
                off = (long)(dh * g.Win + dw) * g.lda + (t - tap * kpt) * PK;
            }
        }
#pragma unroll
        for (int c = 0; c < NRAW; ++c) {
            const float* src = asrc[c] + off;
            if constexpr (TAPS == 9) src = (tapmask[c] >> tap) & 1u ? src : zsrc;
            if constexpr (AREG) {
#if PECLR_X6P_NT & 1
                if (TAPS == 1 && nct == 1) ar[c] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));   // (no other workgroup reads these rows)
                else
#endif
                ar[c] = *reinterpret_cast<const f32x4*>(src);
            }
            else dma16(src, raw_a + c * 1024);
        }
    };
    // rows of the k-step just landed -> three planes [k-half][row][8 k] (this lane: row 16 c + (lane >> 2),
    // k = 4 (lane & 3) ... + 3, i.e. k-half (lane >> 1) & 1, 8-byte slot lane & 1)
    const int poff = ((lane >> 1) & 1) * HALF + (lane >> 2) * 16 + (lane & 1) * 8;
    auto split_store = [&]() {
#pragma unroll
        for (int c = 0; c < NRAW; ++c) {
            f32x4 v;
            if constexpr (AREG) v = ar[c];
            else v = *reinterpret_cast<const f32x4*>(raw + c * 1024 + lane * 16);
            unsigned char* d = planes + poff + c * 256;
            if constexpr (NP == 2) {
                unsigned h[2], l[2];
                split2_pk(v[0], v[1], pair_s, h[0], l[0]);
                split2_pk(v[2], v[3], pair_s, h[1], l[1]);
                *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(l[0], l[1]);
            } else {
                unsigned h[2], m[2], l[2];
                split3_pk(v[0], v[1], h[0], m[0], l[0]);
                split3_pk(v[2], v[3], h[1], m[1], l[1]);
                *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(m[0], m[1]);
                *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(l[0], l[1]);
            }
        }
    };

    // Every VMEM operation of the loop is issued in the MIDDLE of a k-step -- B chunk t + 2 (DMA), then the rows of step
    // t + 2 -- and waited for in the middle of the next one.  The memory counter retires in order, so "the rows have
    // arrived" (the wait hipcc puts in front of their first use when they are register loads; vmcnt(0) by hand when
    // they are DMAs too) implies "my pieces of the older B chunk have landed"; the barrier at the top of step t + 2
    // publishes them.  Three B buffers: t being read, t + 1 landed, t + 2 landing.
    issue_b(0);
    issue_a(0);
    if constexpr (!AREG) PECLR_VMCNT(0);
    split_store();
    if constexpr (!AREG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nk > 1) { issue_b(1); issue_a(1); }

    const int foff = kh * HALF + i * 16;                  // this lane's fragment inside a plane (+ 512 per 32-row tile)
    // one k-step; SPLIT: the rows of step t + 1 are split and stored (t + 1 < nk), interleaved with the first half's MFMAs
    auto kstep = [&](int t, auto split_next) {
        constexpr bool SPLIT = decltype(split_next)::value;
        if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        uint4 af[WM][NP];
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int p = 0; p < NP; ++p) af[a][p] = *reinterpret_cast<const uint4*>(planes + p * PLANE + foff + a * 512);
        const unsigned char* bt = lds + ((ABL & 32) ? 0 : (t % NB)) * CHL + lane * 16;
#pragma unroll
        for (int half = 0; half < NTL / 2; ++half) {
            uint4 bf[2][NP];
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int p = 0; p < NP; ++p) bf[y][p] = *reinterpret_cast<const uint4*>(bt + ((2 * half + y) * NP + p) * 1024);
#define PECLR_X6(P, Q)                                                                        \
    _Pragma("unroll") for (int y = 0; y < 2; ++y) _Pragma("unroll") for (int a = 0; a < WM; ++a) \
        if constexpr (ABL & 8) { asm volatile("" :: "v"(af[a][P].x), "v"(af[a][P].w), "v"(bf[y][Q].x), "v"(bf[y][Q].w)); } \
        else acc[a][2 * half + y] = mman<NP>(af[a][P], bf[y][Q], acc[a][2 * half + y]);
            if constexpr (NP == 3 && !(ABL & 64)) { PECLR_X6(NP - 1, 0) PECLR_X6(0, NP - 1) PECLR_X6(1, 1) }       // (ABL 64: three products only -- timing probe)
            PECLR_X6(1, 0) PECLR_X6(0, 1) PECLR_X6(0, 0)                  // (NP = 2: lo.hi, hi.lo, hi.hi -- smallest first)
#undef PECLR_X6
            if (half == 0 && SPLIT) {
                if constexpr (!AREG) PECLR_VMCNT(0);
                if constexpr (!(ABL & 1)) split_store();       // after this step's fragment reads in program (= LDS) order
                else if constexpr (AREG) { asm volatile("" :: "v"(ar[0]), "v"(ar[NRAW - 1])); }
                if constexpr (AREG && !(ABL & 9) && ILV) {   // (ABL 64: the interleave pattern below still assumes 24 WM products: harmless)
                    // one MFMA, then four of the split's VALU instructions (the two pipes run side by side), a plane store now and then
#pragma unroll
                    for (int q = 0; q < (NP == 3 ? 12 : 6) * WM; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 4 : 6, 0);
                        if (q % 4 == 3 || NP == 2) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                if constexpr (!AREG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (t + 2 < nk) {
                    if constexpr (!(ABL & 4)) issue_b(t + 2);
                    if constexpr (!(ABL & 2)) issue_a(t + 2);
                }
            }
        }
    };
    PECLR_VMCNT(0);                                       // (B chunk 0; for nk > 1 also what was just issued)
    for (int t = 0; t + 1 < nk; ++t) kstep(t, std::true_type{});
    kstep(nk - 1, std::false_type{});
    } else {
        static_assert(!HALO || (TAPS == 9 && AREG && (ABL & ~127) == 0 && !(ABL & 8)), "HALO is the 3x3 / stride-1 path (ABL: 1 no split / plane stores, 2 no patch loads, 4 no B DMA in the loop, 16 no barriers, 32 B fragments from buffer 0, 64 three products)");
        constexpr int NLD = (NPXM * 4 + 255) / 256;      // 16-byte loads per thread and chunk
        constexpr int NDMA = NTL == 4 ? NP : 1;          // B DMA instructions per wave and k-step (at least)
        unsigned char* const patch = lds + RAW0;
        if (tid < 4) reinterpret_cast<unsigned*>(lds + ZOFF)[tid] = 0u;
        const int W = g.W, npx = TM + 2 * W + 2;
        const int wg_m0 = row_block * TM;
        const int nkc = g.lda / PK;
        // this lane's rows as fragment owner: pixel index inside the patch, and which taps lie inside the image
        int fpix[WM];
        unsigned fmask[WM];
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            int row = m0 + a * 32 + i;
            row = row < g.M ? row : g.M - 1;
            const int ow = row % W, oh = (row / W) % g.H;
            fpix[a] = wave * RM + a * 32 + i + W + 1;
            fmask[a] = 0;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int dh = g.flip ? 1 - tap / 3 : tap / 3 - 1, dw = g.flip ? 1 - tap % 3 : tap % 3 - 1;
                if ((unsigned)(oh + dh) < (unsigned)g.H && (unsigned)(ow + dw) < (unsigned)W) fmask[a] |= 1u << tap;
            }
        }
        // patch loader: thread -> (pixel, k-quad) for idx = tid + 256 u; pixels past the patch re-read its last one
        // (pixel tid / 4 + 64 u, k-quad tid & 3; addresses recomputed per chunk: registers are what this variant is short of)
        const int px0 = tid >> 2, kq = tid & 3;
        const int pd0 = (kq >> 1) * PHALF + px0 * 16 + (kq & 1) * 8;
        const float* const pa = g.A + 4 * kq;
        f32x4 pr[NLD];
        auto load_patch = [&](int kc) {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                int px = px0 + 64 * u;
                px = px < npx ? px : npx - 1;
                int r = wg_m0 - W - 1 + px;
                r = r < 0 ? 0 : (r >= g.M ? g.M - 1 : r);
                pr[u] = *reinterpret_cast<const f32x4*>(pa + (size_t)r * g.lda + kc * PK);
            }
        };
        auto store_patch = [&]() {
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                unsigned h[2], m[2], l[2];
                if constexpr (NP == 2) {
                    split2_pk(pr[u][0], pr[u][1], pair_s, h[0], m[0]);
                    split2_pk(pr[u][2], pr[u][3], pair_s, h[1], m[1]);
                } else {
                    split3_pk(pr[u][0], pr[u][1], h[0], m[0], l[0]);
                    split3_pk(pr[u][2], pr[u][3], h[1], m[1], l[1]);
                }
                if (px0 + 64 * u < npx) {
                    unsigned char* d = patch + pd0 + u * 1024;
                    *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
                    *reinterpret_cast<uint2*>(d + PPLANE) = make_uint2(m[0], m[1]);
                    if constexpr (NP == 3) *reinterpret_cast<uint2*>(d + 2 * PPLANE) = make_uint2(l[0], l[1]);
                }
            }
        };
        auto issue_bh = [&](int s) {                      // step s = chunk * 9 + tap reads packed chunk tap * nkc + chunk
            const int kc = s / 9, tap = s - 9 * kc;
            const unsigned char* src = bsrc + (size_t)(tap * nkc + kc) * CHUNK;
            const unsigned d = b_a + (s % NB) * CHL;
            if constexpr (NTL == 4) {
#pragma unroll
                for (int q = 0; q < NP; ++q) dma16(src + (NP * wave_s + q) * 1024, d + (NP * wave_s + q) * 1024);
            } else {
                dma16(src + wave_s * 1024, d + wave_s * 1024);
                if (NP == 3 && wave_s < 2) dma16(src + (4 + wave_s) * 1024, d + (4 + wave_s) * 1024);
            }
        };
        const int nsteps = 9 * nkc;
        load_patch(0);
        asm volatile("" ::: "memory");
        issue_bh(0);
        issue_bh(1);
        store_patch();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the plane stores are in the LDS before the (raw) barrier lets other waves read
        PECLR_VMCNT(0);
        for (int kc = 0; kc < nkc; ++kc) {
            for (int j = 0; j < 9; ++j) {
                const int s = kc * 9 + j;
                // B chunk s has landed: issued after it are chunk s + 1 (and, at j == 1, the next patch's rows)
                if (j == 1 && kc + 1 < nkc) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((ABL & 2) ? NDMA : NDMA + NLD) : "memory");
                else if (s + 1 == nsteps) PECLR_VMCNT(0);        // (nothing was issued after the last chunk)
                else if (s > 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
                if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const int ja = j / 3, jb = j - 3 * ja;
                const int dh = g.flip ? 1 - ja : ja - 1, dw = g.flip ? 1 - jb : jb - 1;
                const int shift = (dh * W + dw) * 16;
                uint4 af[WM][NP];
#pragma unroll
                for (int a = 0; a < WM; ++a) {
                    const bool in = (fmask[a] >> j) & 1u;
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const int off = in ? RAW0 + p * PPLANE + kh * PHALF + fpix[a] * 16 + shift : ZOFF;
                        af[a][p] = *reinterpret_cast<const uint4*>(lds + off);
                    }
                }
                const unsigned char* bt = lds + ((ABL & 32) ? 0 : (s % NB)) * CHL + lane * 16;
#pragma unroll
                for (int half = 0; half < NTL / 2; ++half) {
                    uint4 bf[2][NP];
#pragma unroll
                    for (int y = 0; y < 2; ++y)
#pragma unroll
                        for (int p = 0; p < NP; ++p) bf[y][p] = *reinterpret_cast<const uint4*>(bt + ((2 * half + y) * NP + p) * 1024);
#define PECLR_X6(P, Q)                                                                        \
    _Pragma("unroll") for (int y = 0; y < 2; ++y) _Pragma("unroll") for (int a = 0; a < WM; ++a) \
        acc[a][2 * half + y] = mman<NP>(af[a][P], bf[y][Q], acc[a][2 * half + y]);
                    if constexpr (NP == 3 && !(ABL & 64)) { PECLR_X6(NP - 1, 0) PECLR_X6(0, NP - 1) PECLR_X6(1, 1) }
                    PECLR_X6(1, 0) PECLR_X6(0, 1) PECLR_X6(0, 0)
#undef PECLR_X6
                    if (half == 0) {
                        if (j == 0 && kc + 1 < nkc && !(ABL & 2)) { load_patch(kc + 1); asm volatile("" ::: "memory"); }
                        if (j == 3 && kc + 1 < nkc) {     // the next patch's rows are in (hipcc places its own wait here, long after the issue)
#pragma unroll
                            for (int u = 0; u < NLD; ++u) asm volatile("" :: "v"(pr[u][0]), "v"(pr[u][3]));
                        }
                        if (s + 2 < nsteps && !(ABL & 4)) issue_bh(s + 2);
                    }
                }
            }
            if (kc + 1 < nkc && !(ABL & 1)) {
                if constexpr (!(ABL & 16)) __builtin_amdgcn_s_barrier();             // every wave has read this chunk's patch
                asm volatile("" ::: "memory");
                store_patch();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (published by the barrier at the top of the next step)
            }
        }
    }

    if constexpr (NP == 2) {                              // undo the operands' powers of two (exact)
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
            for (int y = 0; y < NTL; ++y)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][y][r] *= pair_inv;
    }
    // epilogue: wave-private 32 x 32 transposes through LDS (the B buffers, once every wave is done with them), 16 bytes per lane
    __syncthreads();
    float* wlds = reinterpret_cast<float*>(lds + wave * (32 * XEPL * 4));
    if (g.stat_partial) {
        // column statistics of this workgroup's TM x 128 block straight from the accumulators: a lane holds column
        // y * 32 + i for 16 rows per MFMA tile; the two k-halves of a column meet through one cross-lane add, the four
        // waves (different rows, same columns) through 4 KiB of LDS, added in a fixed order
        float* sl = reinterpret_cast<float*>(lds + 4 * (32 * XEPL * 4));          // [wave][2][128]
#pragma unroll
        for (int y = 0; y < NTL; ++y) {
            const float k0 = cst[y * 32 + i];
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int a = 0; a < WM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[a][y][r] - k0;
                    if (m0 + a * 32 + mfma32_row(r, kh) < g.M) { sum += d; sq = fmaf(d, d, sq); }
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
                g.stat_partial[(size_t)((g.M + TM - 1) / TM) * 2 * g.N + n0 + col] = g.stat_shift[n0 + col];
            }
        }
    }
    const int er = lane >> 3, ec = (lane & 7) * 4;
    float* sl = reinterpret_cast<float*>(lds + 4 * (32 * XEPL * 4));              // [wave][2][128] (BatchNorm backward sums)
    // output row of each of this lane's rows (s2d: row (image, i, j) of the class -> input pixel (2 i + ph, 2 j + pw))
    int om[WM][4];
#pragma unroll
    for (int a = 0; a < WM; ++a)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int m = m0 + a * 32 + er + 8 * jj;
            om[a][jj] = m;
            if (s2d && m < g.M) {
                const int j2 = m % g.W, q = m / g.W, i2 = q % g.H, img = q / g.H;
                om[a][jj] = (img * 2 * g.H + 2 * i2 + ph) * (2 * g.W) + 2 * j2 + pw;
            }
        }
#pragma unroll
    for (int y = 0; y < NTL; ++y) {
        const int nt = n0 + y * 32;
        f32x4 bmean, binv, bsc, bsh;
        float sb[4] = {0.f, 0.f, 0.f, 0.f}, sg[4] = {0.f, 0.f, 0.f, 0.f};
        if (g.bb_partial) {
            bmean = *reinterpret_cast<const f32x4*>(cst + 128 + y * 32 + ec);
            binv = *reinterpret_cast<const f32x4*>(cst + 256 + y * 32 + ec);
            bsc = *reinterpret_cast<const f32x4*>(cst + 384 + y * 32 + ec);
            bsh = *reinterpret_cast<const f32x4*>(cst + 512 + y * 32 + ec);
        }
#pragma unroll
        for (int a = 0; a < WM; ++a) {
            const int mt = m0 + a * 32;
            float4 dv[4];
            f32x4 xv[4];
            unsigned mb[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = mt + er + 8 * jj;
                dv[jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (g.addend && m < g.M) {
                    size_t arow = m;
                    bool has = true;
                    if (g.add_h) {
                        const int w = m % g.add_w, hq = m / g.add_w, h = hq % g.add_h, img = hq / g.add_h;
                        has = ((h | w) & 1) == 0;
                        arow = ((size_t)img * (g.add_h >> 1) + (h >> 1)) * (g.add_w >> 1) + (w >> 1);
                    }
                    if (has) {
                        const f32x4* src = reinterpret_cast<const f32x4*>(g.addend + arow * g.ldd + nt + ec);
                        const f32x4 tv = g.stream_out ? __builtin_nontemporal_load(src) : *src;
                        const unsigned ab = g.add_mask ? g.add_mask[(size_t)m * (g.N >> 5) + (nt >> 5)] >> ec : 0xfu;
                        dv[jj] = make_float4(ab & 1u ? tv[0] : 0.f, ab & 2u ? tv[1] : 0.f, ab & 4u ? tv[2] : 0.f, ab & 8u ? tv[3] : 0.f);
                    }
                }
                if (g.bb_partial && m < g.M) {
#if PECLR_X6P_NT & 2
                    xv[jj] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g.bb_x + (size_t)om[a][jj] * g.N + nt + ec));
#else
                    xv[jj] = *reinterpret_cast<const f32x4*>(g.bb_x + (size_t)om[a][jj] * g.N + nt + ec);
#endif
                    mb[jj] = g.bb_mask ? g.bb_mask[(size_t)om[a][jj] * (g.N >> 5) + (nt >> 5)] >> ec : 0u;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) wlds[mfma32_row(r, kh) * XEPL + i] = acc[a][y][r];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int m = mt + er + 8 * jj;
                float4 c = *reinterpret_cast<const float4*>(wlds + (er + 8 * jj) * XEPL + ec);
                c.x += dv[jj].x; c.y += dv[jj].y; c.z += dv[jj].z; c.w += dv[jj].w;
                if (m < g.M) {
                    f32x4* dst = reinterpret_cast<f32x4*>(g.out + (size_t)om[a][jj] * g.ldo + nt + ec);
                    const f32x4 tv = {c.x, c.y, c.z, c.w};
                    if (g.stream_out) __builtin_nontemporal_store(tv, dst);
                    else *dst = tv;
                    if (g.bb_partial) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            bool on = true;
                            if (g.bb_relu) on = g.bb_mask ? (mb[jj] >> q) & 1u : fmaf(xv[jj][q], bsc[q], bsh[q]) > 0.f;
                            const float d = on ? tv[q] : 0.f;
                            sb[q] += d;
                            sg[q] = fmaf(d, (xv[jj][q] - bmean[q]) * binv[q], sg[q]);
                        }
                    }
                }
            }
        }
        if (g.bb_partial) {                               // rows of this wave: lanes with equal (lane & 7) hold the same four columns
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { sb[q] += __shfl_xor(sb[q], o, 64); sg[q] += __shfl_xor(sg[q], o, 64); }
                if (er == 0) { sl[(wave * 2) * 128 + y * 32 + ec + q] = sb[q]; sl[(wave * 2 + 1) * 128 + y * 32 + ec + q] = sg[q]; }
            }
        }
    }
    if (g.bb_partial) {
        __syncthreads();
        const int which = tid >> 7, col = tid & 127;
        if (col < PNL) {
            const float v = ((sl[(0 * 2 + which) * 128 + col] + sl[(1 * 2 + which) * 128 + col]) +
                             sl[(2 * 2 + which) * 128 + col]) + sl[(3 * 2 + which) * 128 + col];
            // (s2d: the four classes' row blocks one after the other)
            const size_t rb = (size_t)row_block + (s2d ? (size_t)blockIdx.y * ((g.M + TM - 1) / TM) : 0);
            g.bb_partial[(rb * 2 + which) * g.N + n0 + col] = v;
        }
    }
}

// ---- weight packing: W[N][K] fp32 (or its transpose) -> fragment-ordered bf16 planes.  One workgroup per
// (128 columns, 16 k) chunk, thread = (column, k-half): 8 k-values -> 16 bytes in each of the chunk's planes.
struct PackDesc {           // device table entry (8 x int64)
    int64_t src, dst;       // fp32 matrix, packed output
    int64_t n, k;           // logical B_t[n][k] extents (n multiple of 128, k multiple of 16)
    int64_t ld;             // leading dimension of src (floats)
    int64_t transposed;     // 0: B_t[n][k] = src[n * ld + k];  1: B_t[n][k] = src[k * ld + n];  T > 1 (a T-tap filter
                            // W[Cout][T][Cin] read for its input gradient, n = ci, k = tap * Cout + co, Cout = k extent / T):
                            // B_t[n][k] = src[((k % Cout) * T + k / Cout) * ld + n]
    int64_t chunk_begin;    // first chunk of this matrix in the launch
    int64_t pad;
};

__global__ __launch_bounds__(256) void x6_pack_kernel(const PackDesc* descs, int count) {
    static_assert(sizeof(PackDesc) == 64 && offsetof(PackDesc, chunk_begin) == 48, "pack_entry_of_chunk reads field 6 of 8 x int64 entries");
    const int d = pack_entry_of_chunk(reinterpret_cast<const int64_t*>(descs), count);
    const PackDesc e = descs[d];
    const int chunk = (int)(blockIdx.x - e.chunk_begin);
    const int nks = (int)(e.k / PK);
    const int ct = chunk / nks, ks = chunk % nks;            // (chunks per matrix: ceil(n / 128) * (k / 16))
    const int col = threadIdx.x & 127, kh = threadIdx.x >> 7;
    const int n = ct * PN + col, k0 = ks * PK + 8 * kh;
    const float* src = reinterpret_cast<const float*>(e.src);
    float v[8];
    if (n >= e.n) {                                           // padding columns of the last chunk (n a multiple of 64 only)
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
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split3_pk(v[2 * q], v[2 * q + 1], h[q], m[q], l[q]);
    unsigned char* dst = reinterpret_cast<unsigned char*>(e.dst) + (size_t)chunk * CHUNK3 + ((col >> 5) * 3) * 1024 +
                         ((col & 31) + 32 * kh) * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(m[0], m[1], m[2], m[3]);
    *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(l[0], l[1], l[2], l[3]);
}

// ---- the same for the fp16 pair (NP = 2): per (128 columns, 16 k) an 8 KiB chunk of eight pieces [32-column block][hi | lo].
// The matrix's power of two needs max |W| over the WHOLE matrix first: x6_absmax_kernel (same grid: one workgroup per chunk,
// one atomic per wave on the matrix's slot -- a maximum does not depend on the order) fills absmax[d]; the pack kernel derives
// the scale from it, applies it, and leaves it in scales[d] for the GEMMs.
__device__ __forceinline__ void pack_load8(const PackDesc& e, int n, int k0, float (&v)[8]) {
    const float* src = reinterpret_cast<const float*>(e.src);
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
}

// max |W| per matrix of a pack table: workgroup (x, d) walks chunks x, x + gridDim.x, ... of matrix d (a matrix and its transpose --
// neighbours in the table, same source -- share the maximum: only the first of the two is read) and ends with ONE conditional atomic.
// (Through round 6 this was the pack kernel's grid, one 8 KiB chunk per workgroup: 127 us for 94 MB.)
__global__ __launch_bounds__(256) void x6_absmax_kernel(const PackDesc* descs, int count, float* absmax) {
    const int d = blockIdx.y;
    const PackDesc e = descs[d];
    if (d > 0 && e.transposed && descs[d - 1].src == e.src && !descs[d - 1].transposed) return;
    const int nks = (int)(e.k / PK), nchunks = (int)((e.n + PN - 1) / PN) * nks;
    const int col = threadIdx.x & 127, kh = threadIdx.x >> 7;
    float m = 0.f;
    if (!e.transposed) {                                      // B_t[n][k] = src[n * ld + k]: rows of k contiguous floats, 16 bytes per lane
        const float* src = reinterpret_cast<const float*>(e.src);
        const int k4 = (int)(e.k / 4);
        const long total = (long)e.n * k4;
        for (long f = (long)blockIdx.x * 256 + threadIdx.x; f < total; f += (long)gridDim.x * 256) {
            const long row = f / k4;
            const float4 v = *reinterpret_cast<const float4*>(src + row * e.ld + 4 * (f - row * k4));
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
    } else
    for (int chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const int ct = chunk / nks, ks = chunk % nks;
        float v[8];
        pack_load8(e, ct * PN + col, ks * PK + 8 * kh, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) m = fmaxf(m, fabsf(v[q]));       // (a NaN weight is not seen here: it still makes its products NaN)
    }
    __shared__ float wmax[4];
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {                                   // (non-negative floats order like their bits; look before the atomic)
        unsigned* p = reinterpret_cast<unsigned*>(absmax + d);
        const unsigned mine = __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3])));
        if (mine > __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(p, mine);
    }
}

// the fp16-pair planes of the matrices of a pack table, each multiplied by the power of two its maximum gives (one workgroup per
// 8 KiB chunk)
__global__ __launch_bounds__(256) void x6_pair_kernel(const PackDesc* descs, int count, const float* absmax, float* scales) {
    static_assert(sizeof(PackDesc) == 64 && offsetof(PackDesc, chunk_begin) == 48, "pack_entry_of_chunk reads field 6 of 8 x int64 entries");
    const int d = pack_entry_of_chunk(reinterpret_cast<const int64_t*>(descs), count);
    const PackDesc e = descs[d];
    // a matrix and its transpose (forward and input-gradient planes of one weight: neighbours in the table, same source) share
    // their maximum: found once, through the first of the two entries
    const int dm = (d > 0 && e.transposed && descs[d - 1].src == e.src && !descs[d - 1].transposed) ? d - 1 : d;
    const int chunk = (int)(blockIdx.x - e.chunk_begin);
    const int nks = (int)(e.k / PK);
    const int ct = chunk / nks, ks = chunk % nks;
    const int col = threadIdx.x & 127, kh = threadIdx.x >> 7;
    float v[8];
    pack_load8(e, ct * PN + col, ks * PK + 8 * kh, v);
    const float sc = pair_scale(absmax[dm]);
    if (chunk == 0 && threadIdx.x == 0) scales[d] = sc;
    unsigned h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2_pk(v[2 * q], v[2 * q + 1], sc, h[q], l[q]);
    unsigned char* dst = reinterpret_cast<unsigned char*>(e.dst) + (size_t)chunk * (8 * 1024) + ((col >> 5) * 2) * 1024 +
                         ((col & 31) + 32 * kh) * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(l[0], l[1], l[2], l[3]);
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int64_t peclr_x6_pack_pair_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % 64 || K % PK) return 0;
    return (int64_t)((N + PN - 1) / PN * PN) * K * 4;
}

// max |W| of every matrix of a descriptor table (the table peclr_x6_pack_f32 takes) -> absmax[count] (zeroed here, then one
// launch over the pack's grid).  First half of the pair-format pack.
extern "C" int peclr_x6_absmax_f32(const void* desc_table, int count, int total_chunks, float* absmax, peclr_stream_t stream) {
    if (!desc_table || !absmax) return PECLR_ERR_NULL;
    if (count <= 0 || total_chunks <= 0) return PECLR_ERR_SHAPE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(absmax, 0, sizeof(float) * (size_t)count, s) != hipSuccess) return launch_status();
    hipLaunchKernelGGL(x6_absmax_kernel, dim3(32, count), dim3(256), 0, s, static_cast<const PackDesc*>(desc_table), count, absmax);
    return launch_status();
}

// Pack the matrices of the table (dst sized by peclr_x6_pack_pair_bytes) as fp16 pairs for the NP = 2 GEMMs, each multiplied by the
// power of two its maximum (absmax[d], from peclr_x6_absmax_f32) gives; scales: float [count], entry d = that power of two -- what
// the GEMM entry points take as peclr_x6_pair.w_scale.
extern "C" int peclr_x6_pack_pair_f32(const void* desc_table, int count, int total_chunks, const float* absmax, float* scales,
                                      peclr_stream_t stream) {
    if (!desc_table || !absmax || !scales) return PECLR_ERR_NULL;
    if (count <= 0 || total_chunks <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(x6_pair_kernel, dim3(total_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PackDesc*>(desc_table), count, absmax, scales);
    return launch_status();
}

extern "C" int64_t peclr_x6_pack_bytes(int N, int K) {
    if (N <= 0 || K <= 0 || N % 64 || K % PK) return 0;      // (N is padded to whole 128-column chunks with zero columns)
    return (int64_t)((N + PN - 1) / PN * PN) * K * 6;
}

extern "C" int peclr_x6_pack_f32(const void* desc_table, int count, int total_chunks, peclr_stream_t stream) {
    if (!desc_table) return PECLR_ERR_NULL;
    if (count <= 0 || total_chunks <= 0) return PECLR_ERR_SHAPE;
    hipLaunchKernelGGL(x6_pack_kernel, dim3(total_chunks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PackDesc*>(desc_table), count);
    return launch_status();
}

extern "C" int peclr_gemm_x6p_tile_rows(int M, int N, int K) {
    // 128-row tiles run three workgroups per CU (768 slots), 256-row tiles two (512 slots) at half the tiles and half the
    // B traffic per flop; pick the one with fewer (rounds of slots) x (rows per tile) -- the workgroup rounds a launch
    // quantises into are what separates the two on ResNet's shapes (tools/exp/gemm_x6p_probe.py) -- and 256 on a tie
    if (M <= 0 || N <= 0 || N % 64) return 0;
    const long nct = N % PN ? N / 64 : N / PN;
    const long t128 = (long)((M + 127) / 128) * nct, t256 = (long)((M + 255) / 256) * nct;
    const long c128 = ((t128 + 767) / 768) * 128, c256 = ((t256 + 511) / 512) * 256;
    return c128 < c256 ? 128 : 256;
}

extern "C" int peclr_gemm_x6p_tile_rows(int M, int N, int K);

static int set_pair(X6PArgs& g, const peclr_x6_pair* pair) {
    g.a_absmax = pair ? pair->a_absmax : nullptr;
    g.w_scale = pair ? pair->w_scale : nullptr;
    return pair && (!pair->a_absmax || !pair->w_scale) ? PECLR_ERR_NULL : PECLR_OK;
}

static void set_bb(X6PArgs& g, const peclr_bn_bwd_fuse* bb) {
    g.bb_x = bb ? bb->x : nullptr; g.bb_mean = bb ? bb->mean : nullptr; g.bb_invstd = bb ? bb->invstd : nullptr;
    g.bb_ss = bb ? bb->scale_shift : nullptr; g.bb_mask = bb ? bb->relu_mask : nullptr; g.bb_relu = bb ? bb->relu : 0;
    g.bb_partial = bb ? bb->partial : nullptr;
}

// outputs larger than the caches are written (and their addends read) with the non-temporal hint (PECLR_X6P_STREAM_OUT=0: never,
// for A/B runs against the memory-side cache)
static bool stream_past_caches(size_t bytes) {
    static const int on = getenv("PECLR_X6P_STREAM_OUT") ? atoi(getenv("PECLR_X6P_STREAM_OUT")) : 1;
    return on && bytes > ((size_t)64 << 20);
}

static int launch_x6p(const X6PArgs& g, int tile_rows, int taps, hipStream_t stream, bool halo = false) {
    const int nrb = (g.M + tile_rows - 1) / tile_rows;
    // 64-column tiles (N a multiple of 64 only) -- and for the entry-gradient GEMMs with K <= 256 (layers 1-3: HBM-bound,
    // their time is the epilogue's addend / BatchNorm-x loads and stores): 37 KiB of LDS and 120 VGPRs put four workgroups on a
    // CU instead of three (420 -> 394 us per launch in the step; the plain 1x1 products of the same shapes lose 5 %: not them)
    const bool narrow = g.N % PN != 0 || (taps == 1 && g.addend != nullptr && g.K <= 256 && g.stride == 1);
    const dim3 grid(8 * ((nrb + 7) / 8) * (narrow ? g.N / 64 : g.N / PN), taps == 9 && g.s2d ? 4 : 1);
// (ILV: A/B'd again in round 5 -- with it hipcc groups each accumulator's six products and puts the split behind them, without it
// products and split alternate; every tag of the step within 1 % either way: the waves sharing a SIMD hide a wave's own order)
#ifndef PECLR_X6P_ILV
#define PECLR_X6P_ILV true
#endif
    const bool pair = g.a_absmax != nullptr;              // fp16-pair arithmetic (NP = 2): Bp holds pair planes
#define PECLR_LAUNCH(WM_, TAPS_)                                                                                      \
    do {                                                                                                              \
        if (pair) {                                                                                                   \
            if (narrow) hipLaunchKernelGGL((gemm_x6p_kernel<WM_, 0, true, PECLR_X6P_ILV, TAPS_, 2, false, 2>), grid, dim3(256), 0, stream, g); \
            else hipLaunchKernelGGL((gemm_x6p_kernel<WM_, 0, true, PECLR_X6P_ILV, TAPS_, 4, false, 2>), grid, dim3(256), 0, stream, g);       \
        } else if (narrow) hipLaunchKernelGGL((gemm_x6p_kernel<WM_, 0, true, PECLR_X6P_ILV, TAPS_, 2>), grid, dim3(256), 0, stream, g); \
        else hipLaunchKernelGGL((gemm_x6p_kernel<WM_, 0, true, PECLR_X6P_ILV, TAPS_, 4>), grid, dim3(256), 0, stream, g);       \
    } while (0)
#define PECLR_LAUNCH_HALO(WM_, NP_)                                                                                   \
    do {                                                                                                              \
        if (narrow) hipLaunchKernelGGL((gemm_x6p_kernel<WM_, 0, true, true, 9, 2, true, NP_>), grid, dim3(256), 0, stream, g); \
        else hipLaunchKernelGGL((gemm_x6p_kernel<WM_, 0, true, true, 9, 4, true, NP_>), grid, dim3(256), 0, stream, g);       \
    } while (0)
    if (taps == 9 && halo) {
        if (tile_rows == 256) { if (pair) PECLR_LAUNCH_HALO(2, 2); else PECLR_LAUNCH_HALO(2, 3); }
        else { if (pair) PECLR_LAUNCH_HALO(1, 2); else PECLR_LAUNCH_HALO(1, 3); }
    } else if (taps == 9) {
        if (tile_rows == 256) PECLR_LAUNCH(2, 9); else PECLR_LAUNCH(1, 9);
    } else {
        if (tile_rows == 256) PECLR_LAUNCH(2, 1); else PECLR_LAUNCH(1, 1);
    }
#undef PECLR_LAUNCH
#undef PECLR_LAUNCH_HALO
    return launch_status();
}

extern "C" int peclr_conv3x3_s2_dgrad_x6p_f32(int NB, int Ho, int Wo, int Cout, int Cin, const float* dY, const void* Bp, float* dX,
                                              int tile_rows, const float* zeros, const peclr_bn_bwd_fuse* bb, const peclr_x6_pair* pair,
                                              peclr_stream_t stream) {
    if (!dY || !Bp || !dX || !zeros) return PECLR_ERR_NULL;
    if (bb && (!bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial)) return PECLR_ERR_NULL;
    if (NB <= 0 || Ho <= 0 || Wo <= 0 || Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % PK) return PECLR_ERR_SHAPE;
    if ((long)NB * Ho * Wo * 4 > 0x7fffffffL) return PECLR_ERR_SHAPE;
    if (!aligned16(dY) || !aligned16(Bp) || !aligned16(dX) || !aligned16(zeros)) return PECLR_ERR_ALIGN;
    const int M = NB * Ho * Wo;                           // rows of one parity class
    if (tile_rows == 0) tile_rows = peclr_gemm_x6p_tile_rows(M, Cin, 4 * Cout);
    if (tile_rows != 128 && tile_rows != 256) return PECLR_ERR_UNSUPPORTED;
    X6PArgs g;
    g.A = dY; g.Bp = Bp; g.addend = nullptr; g.out = dX;
    g.M = M; g.N = Cin; g.K = 9 * Cout; g.lda = Cout; g.ldo = Cin; g.ldd = Cin; g.add_h = g.add_w = 0; g.add_mask = nullptr;
    g.stream_out = stream_past_caches((size_t)M * 4 * Cin * sizeof(float));
    g.stat_shift = nullptr; g.stat_partial = nullptr;
    g.H = Ho; g.W = Wo; g.flip = 1; g.zeros = zeros; g.stride = 1; g.Hin = Ho; g.Win = Wo; g.s2d = 1;
    set_bb(g, bb);
    if (set_pair(g, pair)) return PECLR_ERR_NULL;
    return launch_x6p(g, tile_rows, 9, static_cast<hipStream_t>(stream));
}

extern "C" int peclr_conv_s2_x6p_f32(int NB, int H, int W, int Cin, int Cout, int taps, const float* X, const void* Bp, float* Y,
                                     int tile_rows, const float* zeros, const float* stat_shift, float* stat_partial,
                                     const peclr_x6_pair* pair, peclr_stream_t stream) {
    if (!X || !Bp || !Y || !zeros || (stat_partial && !stat_shift)) return PECLR_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0 || H % 2 || W % 2 || Cin <= 0 || Cout <= 0 || Cout % 64 || Cin % PK || (taps != 1 && taps != 9))
        return PECLR_ERR_SHAPE;
    if (!aligned16(X) || !aligned16(Bp) || !aligned16(Y) || !aligned16(zeros)) return PECLR_ERR_ALIGN;
    const int Ho = H / 2, Wo = W / 2, M = NB * Ho * Wo;
    if (tile_rows == 0) tile_rows = peclr_gemm_x6p_tile_rows(M, Cout, taps * Cin);
    if (tile_rows != 128 && tile_rows != 256) return PECLR_ERR_UNSUPPORTED;
    X6PArgs g;
    g.A = X; g.Bp = Bp; g.addend = nullptr; g.out = Y;
    g.M = M; g.N = Cout; g.K = taps * Cin; g.lda = Cin; g.ldo = Cout; g.ldd = Cout; g.add_h = g.add_w = 0; g.add_mask = nullptr;
    g.stream_out = stream_past_caches((size_t)M * Cout * sizeof(float));
    g.stat_shift = stat_shift; g.stat_partial = stat_partial;
    g.H = Ho; g.W = Wo; g.flip = 0; g.zeros = zeros; g.stride = 2; g.Hin = H; g.Win = W; g.s2d = 0;
    set_bb(g, nullptr);
    if (set_pair(g, pair)) return PECLR_ERR_NULL;
    return launch_x6p(g, tile_rows, taps, static_cast<hipStream_t>(stream));
}

extern "C" int peclr_conv3x3_x6p_f32(int NB, int H, int W, int Cin, int Cout, const float* X, const void* Bp, float* Y,
                                     const float* addend, int flip, int tile_rows, int variant, const float* zeros,
                                     const float* stat_shift, float* stat_partial, const peclr_bn_bwd_fuse* bb,
                                     const peclr_x6_pair* pair, peclr_stream_t stream) {
    if (!X || !Bp || !Y || !zeros || (stat_partial && !stat_shift)) return PECLR_ERR_NULL;
    if (variant != 0 && variant != 1) return PECLR_ERR_UNSUPPORTED;
    if (bb && (stat_partial || !bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial || Cout % 32)) return PECLR_ERR_NULL;
    if (NB <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cout % 64 || Cin % PK) return PECLR_ERR_SHAPE;
    if ((long)NB * H * W > 0x7FFFFFFFL / 2) return PECLR_ERR_SHAPE;
    if (!aligned16(X) || !aligned16(Bp) || !aligned16(Y) || !aligned16(zeros) || (addend && !aligned16(addend))) return PECLR_ERR_ALIGN;
    const int M = NB * H * W;
    if (tile_rows == 0) tile_rows = peclr_gemm_x6p_tile_rows(M, Cout, 9 * Cin);
    if (tile_rows != 128 && tile_rows != 256) return PECLR_ERR_UNSUPPORTED;
    X6PArgs g;
    g.A = X; g.Bp = Bp; g.addend = addend; g.out = Y;
    g.M = M; g.N = Cout; g.K = 9 * Cin; g.lda = Cin; g.ldo = Cout; g.ldd = Cout; g.add_h = g.add_w = 0; g.add_mask = nullptr;
    g.stream_out = stream_past_caches((size_t)M * Cout * sizeof(float));
    g.stat_shift = stat_shift; g.stat_partial = stat_partial;
    g.H = H; g.W = W; g.flip = flip ? 1 : 0; g.zeros = zeros; g.stride = 1; g.Hin = H; g.Win = W; g.s2d = 0;
    set_bb(g, bb);
    if (set_pair(g, pair)) return PECLR_ERR_NULL;
    return launch_x6p(g, tile_rows, 9, static_cast<hipStream_t>(stream), variant == 1 && W <= 62 && Cin >= 2 * PK);
}

static int gemm_x6p_host(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc, int add_h, int add_w,
                         const unsigned* add_mask,
                                  const float* addend, int ldd, int tile_rows, const float* stat_shift, float* stat_partial,
                                  const peclr_bn_bwd_fuse* bb, const peclr_x6_pair* pair, peclr_stream_t stream) {
    if (!A || !Bp || !C || (stat_partial && !stat_shift)) return PECLR_ERR_NULL;
    if (bb && (stat_partial || !bb->x || !bb->mean || !bb->invstd || !bb->scale_shift || !bb->partial || ldc != N)) return PECLR_ERR_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || N % 64 || K % PK) return PECLR_ERR_SHAPE;
    if (add_h && (!addend || add_h < 2 || add_w < 2 || add_h % 2 || add_w % 2 || M % (add_h * add_w))) return PECLR_ERR_SHAPE;
    if (lda % 4 || lda < K || ldc % 4 || ldc < N || (addend && (ldd % 4 || ldd < N))) return PECLR_ERR_SHAPE;
    if (!aligned16(A) || !aligned16(Bp) || !aligned16(C) || (addend && !aligned16(addend))) return PECLR_ERR_ALIGN;
    if (tile_rows == 0) tile_rows = peclr_gemm_x6p_tile_rows(M, N, K);
    if (tile_rows != 128 && tile_rows != 256) return PECLR_ERR_UNSUPPORTED;
    X6PArgs g;
    g.A = A; g.Bp = Bp; g.addend = addend; g.out = C;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldo = ldc; g.ldd = ldd;
    g.add_h = add_h; g.add_w = add_w; g.add_mask = add_mask;
    g.stream_out = stream_past_caches((size_t)M * N * sizeof(float));
    g.stat_shift = stat_shift; g.stat_partial = stat_partial;
    g.H = g.W = 1; g.flip = 0; g.zeros = nullptr; g.stride = 1; g.Hin = g.Win = 1; g.s2d = 0;
    set_bb(g, bb);
    if (set_pair(g, pair)) return PECLR_ERR_NULL;
    return launch_x6p(g, tile_rows, 1, static_cast<hipStream_t>(stream));
}

extern "C" int peclr_gemm_x6p_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                                  const float* addend, int ldd, int tile_rows, const float* stat_shift, float* stat_partial,
                                  const peclr_bn_bwd_fuse* bb, const peclr_x6_pair* pair, peclr_stream_t stream) {
    return gemm_x6p_host(M, N, K, A, lda, Bp, C, ldc, 0, 0, nullptr, addend, ldd, tile_rows, stat_shift, stat_partial, bb, pair, stream);
}

extern "C" int peclr_gemm_x6p_s2add_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                                        const float* addend_half, int ldd, int H, int W, int tile_rows,
                                        const peclr_bn_bwd_fuse* bb, const peclr_x6_pair* pair, peclr_stream_t stream) {
    if (!addend_half || H <= 0 || W <= 0) return PECLR_ERR_NULL;
    return gemm_x6p_host(M, N, K, A, lda, Bp, C, ldc, H, W, nullptr, addend_half, ldd, tile_rows, nullptr, nullptr, bb, pair, stream);
}

extern "C" int peclr_gemm_x6p_maskadd_f32(int M, int N, int K, const float* A, int lda, const void* Bp, float* C, int ldc,
                                          const float* addend, int ldd, const unsigned* addend_mask, int tile_rows,
                                          const peclr_bn_bwd_fuse* bb, const peclr_x6_pair* pair, peclr_stream_t stream) {
    if (!addend || !addend_mask) return PECLR_ERR_NULL;
    if (N % 32 || ((size_t)addend_mask & 3)) return PECLR_ERR_SHAPE;
    return gemm_x6p_host(M, N, K, A, lda, Bp, C, ldc, 0, 0, addend_mask, addend, ldd, tile_rows, nullptr, nullptr, bb, pair, stream);
}
