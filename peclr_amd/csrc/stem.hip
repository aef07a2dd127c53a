// The encoder's stem: 7x7 / stride-2 / padding-3 convolution of the fp32 input images (3 channels) to 64 channels
// (torchvision ResNet `conv1`, built by /root/reference/src/models/resnet_model.py:15) as a direct convolution on the bf16 /
// fp16 matrix cores -- the last forward convolution of the step that ran on MIOpen (fp32: 783 us, its fp32-MFMA implicit GEMM;
// bf16: 373 us + a cast pass of the images).
//
// fp32 runs: the exact three-way bf16 split of both operands, six products, fp32 accumulation (as gemm_x6p.hip: error class of
// an fp32 kernel).  16-bit (autocast) runs: the fp32 images are rounded to bf16 / fp16 while they are staged (autocast's cast of
// the input rides here) and one product is taken.
//
// A 3-channel NHWC image has 12-byte pixels; staged into LDS every pixel gets a fourth, zero channel: 8 bytes per pixel and
// plane.  K is ordered (kh, kw, c4): a filter row is 7 x 4 = 28 -> 32 k = four slices of 8 (pixel pairs); 28 slices = 14 steps of
// 16 k.  The A fragment of output pixel `ow` for slice (kh, s) is then the 16 bytes at pixel 2 ow + 2 s of patch row kh --
// one aligned ds_read_b128, lane-linear (output pixels of a wave are consecutive: stride 16 B), conflict-free.  The padding
// k (kw = 7, c = 3) meets zero weights.
//
// Workgroup: two output rows x 128 output pixels (four waves of 32 pixels x 2 rows) x all 64 channels; it stages the nine
// input rows x 261 pixels its taps read ONCE (split in registers, three planes), streams the packed filter (6 KiB per step, from
// L2) by LDS-DMA through three stages, and writes NHWC rows of 64 channels through wave-private transposes (16-byte stores).
// Optional epilogue: the training statistics of the BatchNorm that follows (peclr_bn2d_stats' partial layout), so that the
// one statistics pass left in the step goes too.
#include <stdlib.h>

#include <type_traits>

#include "common.hpp"

namespace peclr {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));

constexpr int SK = 14;                   // k-steps of 16 (28 slices of 8: kh = slice >> 2, pixel pair = slice & 3)
constexpr int TPX = 128;                 // output pixels of a row per workgroup
constexpr int PWP = 2 * TPX + 6;         // staged pixels per patch row (261 used, the last is read by the zero-weighted k; 16-byte pitch)
constexpr int PROW = PWP * 8;            // bytes per patch row and plane
constexpr int NROW = 9;                  // input rows under two output rows
constexpr int XE = 68;                   // floats per row of the epilogue's 32 x 64 transposes

struct StemArgs {
    const float* x;                      // [N][Hin][Win][3]
    const void* planes;                  // packed filter (peclr_stem_pack)
    void* y;                             // [N][Ho][Wo][64] fp32 / bf16 / fp16
    int N, Hin, Win, Ho, Wo;
    int tiles_w;                         // workgroups per output row pair
    const float* stat_shift;             // optional: [64]
    float* stat_partial;                 // [workgroups][2][64] + the shift row
};

struct X6 {                              // fp32: three bf16 planes, six products
    static constexpr int NP = 3;
    typedef float Out;
    static __device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
};
struct HB {                              // bf16 autocast
    static constexpr int NP = 1;
    typedef uint16_t Out;
    static __device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) { return pk_bf16(a, b); }
    static __device__ __forceinline__ float up(unsigned lo16) { return __uint_as_float(lo16 << 16); }
};
struct HF {                              // fp16 autocast (the reference's precision: 16)
    static constexpr int NP = 1;
    typedef uint16_t Out;
    static __device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, f32x16 acc) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        const f32x2_t v = {a, b};
        return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
    }
    static __device__ __forceinline__ float up(unsigned lo16) { return (float)__builtin_bit_cast(_Float16, (unsigned short)lo16); }
};

__device__ __forceinline__ void sdma16(const void* src, unsigned lds_byte_offset) {     // see gemm_x6p.hip `dma16`
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :: "v"(src), "s"(lds_byte_offset) : "memory", "m0");
}

template <typename F>
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(StemArgs g) {
    constexpr int NP = F::NP;
    constexpr int PLANE = NROW * PROW;                   // 19 008 bytes
    constexpr int CH = NP * 2 * 1024;                    // bytes of packed filter per k-step
    constexpr int NB = 4;                                // filter stages: three steps of DMA run-ahead (a step is 768 MFMA clocks, an L2 round trip ~1 500)
    constexpr int B0 = 0, P0 = NB * CH;                  // filter stages first (LDS-DMA targets below 64 KiB), then the patch
    constexpr int EPI = 4 * 32 * XE * 4 + 4 * 2 * 64 * 4;
    constexpr int TOTAL = P0 + NP * PLANE > EPI ? P0 + NP * PLANE : EPI;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[TOTAL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, half = lane >> 5;
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(__UINTPTR_TYPE__)(lptr_t)lds);
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);

    // workgroup -> (image, output row pair, 128-pixel tile); consecutive workgroups walk along a row pair, then down the image
    const int per_img = ((g.Ho + 1) >> 1) * g.tiles_w;
    const int img = blockIdx.x / per_img, rem = blockIdx.x - img * per_img;
    const int rp = rem / g.tiles_w, cb = rem - rp * g.tiles_w;
    const int oh0 = 2 * rp, ow0 = cb * TPX;
    const int ih0 = 2 * oh0 - 3, iw0 = 2 * ow0 - 3;

    // ---- packed filter: this wave's pieces of step t -> stage t % NB (2 NP pieces of 1 KiB per step)
    const unsigned char* bsrc = static_cast<const unsigned char*>(g.planes) + lane * 16;
    auto issue_b = [&](int t) {
        const unsigned char* s = bsrc + (size_t)t * CH;
        const unsigned d = lds0 + B0 + (t % NB) * CH;
        if constexpr (NP == 3) {                          // six pieces: waves 0, 1 two each, waves 2, 3 one
            sdma16(s + wave_s * 1024, d + wave_s * 1024);
            if (wave_s < 2) sdma16(s + (4 + wave_s) * 1024, d + (4 + wave_s) * 1024);
        } else if (wave_s < 2) {
            sdma16(s + wave_s * 1024, d + wave_s * 1024);
        }
    };

    // the first filter chunks travel while the patch is being staged (their targets are not the patch's)
    issue_b(0);
    issue_b(1);
    issue_b(2);
    // ---- the patch: nine input rows x PWP pixels, fp32 -> NP planes of (r, g, b, 0) 16-bit quadruples
    constexpr int NSLOT = NROW * PWP;                    // 2 358 pixel slots
    constexpr int NLD = (NSLOT + 255) / 256;             // 10 per thread
    struct __attribute__((packed, aligned(4))) Rgb { float r, g, b; };       // one 12-byte load per pixel (4-byte aligned)
    Rgb px[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int idx = tid + 256 * u;
        const int r = idx / PWP, p = idx - r * PWP;
        const int ih = ih0 + r, iw = iw0 + p;
        const bool in = idx < NSLOT && p < 2 * TPX + 5 && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
        const Rgb* src = reinterpret_cast<const Rgb*>(g.x + (((size_t)img * g.Hin + (in ? ih : 0)) * g.Win + (in ? iw : 0)) * 3);
        px[u] = in ? *src : Rgb{0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int idx = tid + 256 * u;
        if (idx < NSLOT) {
            unsigned char* d = lds + P0 + idx * 8;       // (row r, pixel p) = slot idx: rows are PWP slots apart
            if constexpr (NP == 3) {
                unsigned h[2], m[2], l[2];
                split3_pk(px[u].r, px[u].g, h[0], m[0], l[0]);
                split3_pk(px[u].b, 0.f, h[1], m[1], l[1]);
                *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
                *reinterpret_cast<uint2*>(d + PLANE) = make_uint2(m[0], m[1]);
                *reinterpret_cast<uint2*>(d + 2 * PLANE) = make_uint2(l[0], l[1]);
            } else {
                *reinterpret_cast<uint2*>(d) = make_uint2(F::pack2(px[u].r, px[u].g), F::pack2(px[u].b, 0.f));
            }
        }
    }
    // every global load of the patch has been consumed (the stores above needed the data -- the compiler's own waits, which
    // retire the older filter requests with them): from here on the only memory operations in flight are filter DMAs,
    // waited for by hand

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][y][r] = 0.f;

    // this lane's fragment of (row a, step t): slice 2 t + half -> patch row 2 a + (slice >> 2), pixel 2 (32 wave + i) + 2 (slice & 3)
    const int fbase = P0 + (2 * (32 * wave + i)) * 8;
    for (int t = 0; t < SK; ++t) {
        // filter chunk t has landed: younger than it are at most chunks t + 1 and t + 2 (this wave's pieces: 2 / 1 / 0 requests
        // per chunk; the last two steps simply wait for everything)
        if (t + 2 < SK) {
            if (NP == 3 && wave_s < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (NP == 3 || wave_s < 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (t == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the patch stores are in the LDS before the barrier publishes them
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + 3 < SK) issue_b(t + 3);                   // into the stage step t - 1 read
        const int slice = 2 * t + half;
        const int kh = slice >> 2, s = slice & 3;
        uint4 af[2][NP], bf[2][NP];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int p = 0; p < NP; ++p)
                af[a][p] = *reinterpret_cast<const uint4*>(lds + fbase + p * PLANE + (2 * a + kh) * PROW + s * 16);
        if (s == 3) {
            // the pair's second pixel is the padding tap kw = 7: a REAL neighbouring pixel under a zero weight.  Zero the operand too,
            // so that a non-finite pixel there cannot reach this output through 0 * inf (a true 7x7 convolution never reads it)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int p = 0; p < NP; ++p) af[a][p].z = af[a][p].w = 0u;
        }
        const unsigned char* bt = lds + B0 + (t % NB) * CH + lane * 16;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int p = 0; p < NP; ++p) bf[y][p] = *reinterpret_cast<const uint4*>(bt + (y * NP + p) * 1024);
        if constexpr (NP == 3) {
#define PECLR_S6(P, Q)                                                                             \
    _Pragma("unroll") for (int y = 0; y < 2; ++y) _Pragma("unroll") for (int a = 0; a < 2; ++a)    \
        acc[a][y] = F::mma(af[a][P], bf[y][Q], acc[a][y]);
            PECLR_S6(2, 0) PECLR_S6(0, 2) PECLR_S6(1, 1) PECLR_S6(1, 0) PECLR_S6(0, 1) PECLR_S6(0, 0)
#undef PECLR_S6
        } else {
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int a = 0; a < 2; ++a) acc[a][y] = F::mma(af[a][0], bf[y][0], acc[a][y]);
        }
    }
    __syncthreads();                                      // every wave is done with the patch and the stages: the epilogue re-uses them

    // ---- epilogue.  16-bit outputs: the stored value is the rounded accumulator, and the statistics are those of the rounded
    // values (what a pass over the stored tensor would read)
    if constexpr (NP == 1) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int y = 0; y < 2; ++y)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const unsigned p = F::pack2(acc[a][y][r], acc[a][y][r + 1]);
                    acc[a][y][r] = F::up(p & 0xFFFFu);
                    acc[a][y][r + 1] = F::up(p >> 16);
                }
    }
    const int wpx0 = ow0 + 32 * wave;                     // first output pixel of this wave
    if (g.stat_partial) {
        float* sl = reinterpret_cast<float*>(lds + 4 * 32 * XE * 4);      // [wave][2][64]
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const float k0 = g.stat_shift[y * 32 + i];
            float sum = 0.f, sq = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[a][y][r] - k0;
                    if (oh0 + a < g.Ho && wpx0 + mfma32_row(r, half) < g.Wo) { sum += d; sq = fmaf(d, d, sq); }
                }
            sum += __shfl_xor(sum, 32, 64);
            sq += __shfl_xor(sq, 32, 64);
            if (half == 0) { sl[(wave * 2) * 64 + y * 32 + i] = sum; sl[(wave * 2 + 1) * 64 + y * 32 + i] = sq; }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, col = tid & 63;
            const float v = ((sl[(0 * 2 + which) * 64 + col] + sl[(1 * 2 + which) * 64 + col]) + sl[(2 * 2 + which) * 64 + col]) +
                            sl[(3 * 2 + which) * 64 + col];
            g.stat_partial[((size_t)blockIdx.x * 2 + which) * 64 + col] = v;
            if (blockIdx.x == 0 && which == 0) g.stat_partial[(size_t)gridDim.x * 2 * 64 + col] = g.stat_shift[col];
        }
    }
    // wave-private 32 x 64 transposes: a lane then owns 8 consecutive channels of a pixel
    float* wl = reinterpret_cast<float*>(lds + wave * (32 * XE * 4));
    const int er = lane >> 3, ec = (lane & 7) * 8;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oh = oh0 + a;
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 16; ++r) wl[mfma32_row(r, half) * XE + y * 32 + i] = acc[a][y][r];
        // (a wave's own stores and loads of its own region: in order, no barrier)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ow = wpx0 + er + 8 * jj;
            const float4 c0 = *reinterpret_cast<const float4*>(wl + (er + 8 * jj) * XE + ec);
            const float4 c1 = *reinterpret_cast<const float4*>(wl + (er + 8 * jj) * XE + ec + 4);
            if (oh < g.Ho && ow < g.Wo) {
                const size_t o = (((size_t)img * g.Ho + oh) * g.Wo + ow) * 64 + ec;
                if constexpr (NP == 3) {
                    float* out = static_cast<float*>(g.y) + o;
                    *reinterpret_cast<float4*>(out) = c0;
                    *reinterpret_cast<float4*>(out + 4) = c1;
                } else {
                    uint16_t* out = static_cast<uint16_t*>(g.y) + o;
                    *reinterpret_cast<uint4*>(out) = make_uint4(F::pack2(c0.x, c0.y), F::pack2(c0.z, c0.w), F::pack2(c1.x, c1.y), F::pack2(c1.z, c1.w));
                }
            }
        }
    }
}

// ---- filter packing: W[64][3][7][7] (any strides) -> fragment order.  Step t, column tile y, plane p: 1 KiB piece, lane l's 16
// bytes = column 32 y + (l & 31), slice 2 t + (l >> 5): eight k = (kw = 2 s + (kk >> 2), c = kk & 3), zero where kw = 7 or c = 3.
__global__ __launch_bounds__(128) void stem_pack_kernel(const float* w, long sn, long sc, long sh, long sw, unsigned char* planes, int fmt) {
    const int t = blockIdx.x, y = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = 32 * y + (l & 31), slice = 2 * t + (l >> 5), kh = slice >> 2, s = slice & 3;
    float v[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        const int kw = 2 * s + (kk >> 2), c = kk & 3;
        v[kk] = (kw < 7 && c < 3) ? w[n * sn + c * sc + kh * sh + kw * sw] : 0.f;
    }
    if (fmt == 0) {
        unsigned h[4], m[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) split3_pk(v[2 * q], v[2 * q + 1], h[q], m[q], lo[q]);
        unsigned char* d = planes + (size_t)t * 6144 + (y * 3) * 1024 + l * 16;
        *reinterpret_cast<uint4*>(d) = make_uint4(h[0], h[1], h[2], h[3]);
        *reinterpret_cast<uint4*>(d + 1024) = make_uint4(m[0], m[1], m[2], m[3]);
        *reinterpret_cast<uint4*>(d + 2048) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    } else {
        unsigned q4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) q4[q] = fmt == 2 ? HF::pack2(v[2 * q], v[2 * q + 1]) : HB::pack2(v[2 * q], v[2 * q + 1]);
        *reinterpret_cast<uint4*>(planes + (size_t)t * 2048 + y * 1024 + l * 16) = make_uint4(q4[0], q4[1], q4[2], q4[3]);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the stem: dW[n][kh][kw][c] = sum over (image, oh, ow) of dY[oh][ow][n] . X[2 oh - 3 + kh][2 ow - 3 + kw][c]
// -- MIOpen's fp32 kernel for it (1.0 ms) was the last MIOpen launch of the fp32 step.  On the matrix cores the contraction index
// is the OUTPUT PIXEL: C[n][(kh, kw, c4)] = sum_p A[n][p] . B[p][(kh, kw, c4)], 64 x 224 outputs (kw = 7 and c = 3 columns are
// discarded), sixteen pixels per MFMA.  Both fragments want eight consecutive pixels per lane:
//   * A = dY^T: a tile's 64 pixels x 64 channels are split (fp32) / taken as they are (16-bit) and stored TRANSPOSED into LDS,
//     [plane][channel][pixel] with a 144-byte channel pitch (2-byte stores, a wave writes 128 contiguous bytes; the 16-byte
//     fragment reads of 32 channels hit 32 distinct bank groups);
//   * B = the image patch as the forward kernel stages it ([plane][row][pixel][r g b 0], 8 bytes per pixel).  The element of
//     column (kw, c) for output pixel `ow` sits at pixel 2 ow + kw: consecutive output pixels are 16 bytes apart, so a lane
//     gathers its eight k with eight ds_read_u16 at immediate offsets 16 e -- the 32 columns of a tile are 64
//     contiguous bytes, the reads are conflict-free.
// Workgroup: persistent, walks tiles of 64 output pixels of one output row (seven input rows x 134 pixels of patch), two LDS
// stages (the next tile's global loads fly during the products, its split / stores follow them), one barrier per tile; wave w
// owns filter rows kh = 2 w, 2 w + 1 (wave 3: kh = 6) x all 64 channels: four 32 x 32 accumulators; at the end one fp32 slab
// [64][224] per workgroup, summed in a fixed order by peclr_slab_reduce_f32 (deterministic).
constexpr int WPX = 64;                   // output pixels per tile
constexpr int WPW = 2 * WPX + 6;          // patch pixels per row (133 used)
constexpr int WROW = WPW * 8;             // 1 072 bytes
constexpr int WAP = (WPX + 8) * 2;        // bytes per channel row of the transposed dY planes (144)

struct StemWArgs {
    const float* x;                       // [N][Hin][Win][3] fp32 images
    const void* dy;                       // [N][Ho][Wo][64] fp32 / bf16 / fp16
    float* slabs;                         // [gridDim.x][64][224]
    int N, Hin, Win, Ho, Wo, tiles_w;
    long long tiles;
    int abl;                              // experiments (PECLR_STEM_WGRAD_ABL): 1 no dY loads, 2 no dY stores, 4 no gathers, 8 no products, 16 no patch, 32 dY loads non-temporal
};

// eight 16-bit values at addr + OFF + 16 e, each zero-extended into its own register (gfx950 runs with SRAM-ECC: its d16 loads
// do NOT preserve the other half of the destination, so ds_read_u16_d16 / _d16_hi pairs cannot assemble a word in place; the
// halves are merged by one v_lshl_or_b32 per pair after the wait)
struct Gather {
    unsigned lo[4], hi[4];                               // k = 2 q and k = 2 q + 1
};
template <int OFF>
__device__ __forceinline__ void gather8(unsigned addr, Gather& r) {
    asm volatile(
        "ds_read_u16 %0, %8 offset:%9\n\t"
        "ds_read_u16 %4, %8 offset:%10\n\t"
        "ds_read_u16 %1, %8 offset:%11\n\t"
        "ds_read_u16 %5, %8 offset:%12\n\t"
        "ds_read_u16 %2, %8 offset:%13\n\t"
        "ds_read_u16 %6, %8 offset:%14\n\t"
        "ds_read_u16 %3, %8 offset:%15\n\t"
        "ds_read_u16 %7, %8 offset:%16"
        : "=v"(r.lo[0]), "=v"(r.lo[1]), "=v"(r.lo[2]), "=v"(r.lo[3]), "=v"(r.hi[0]), "=v"(r.hi[1]), "=v"(r.hi[2]), "=v"(r.hi[3])
        : "v"(addr), "n"(OFF), "n"(OFF + 16), "n"(OFF + 32), "n"(OFF + 48), "n"(OFF + 64), "n"(OFF + 80), "n"(OFF + 96), "n"(OFF + 112)
        : "memory");
}
// "the gathers of these fragments have landed" -- tied to their registers, so that nothing using them is scheduled above the wait
// (the compiler does not know that the statements above read the LDS asynchronously)
template <int NP>
__device__ __forceinline__ void gather_fence(Gather (&g)[NP]) {
    if constexpr (NP == 3) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g[0].lo[0]), "+v"(g[0].lo[1]), "+v"(g[0].lo[2]), "+v"(g[0].lo[3]), "+v"(g[0].hi[0]), "+v"(g[0].hi[1]),
                       "+v"(g[0].hi[2]), "+v"(g[0].hi[3]), "+v"(g[1].lo[0]), "+v"(g[1].lo[1]), "+v"(g[1].lo[2]), "+v"(g[1].lo[3]),
                       "+v"(g[1].hi[0]), "+v"(g[1].hi[1]), "+v"(g[1].hi[2]), "+v"(g[1].hi[3]), "+v"(g[2].lo[0]), "+v"(g[2].lo[1]),
                       "+v"(g[2].lo[2]), "+v"(g[2].lo[3]), "+v"(g[2].hi[0]), "+v"(g[2].hi[1]), "+v"(g[2].hi[2]), "+v"(g[2].hi[3])
                     :: "memory");
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(g[0].lo[0]), "+v"(g[0].lo[1]), "+v"(g[0].lo[2]), "+v"(g[0].lo[3]), "+v"(g[0].hi[0]), "+v"(g[0].hi[1]),
                       "+v"(g[0].hi[2]), "+v"(g[0].hi[3])
                     :: "memory");
    }
}
__device__ __forceinline__ uint4 merge(const Gather& g) {
    return make_uint4(g.lo[0] | (g.hi[0] << 16), g.lo[1] | (g.hi[1] << 16), g.lo[2] | (g.hi[2] << 16), g.lo[3] | (g.hi[3] << 16));
}

template <typename F>
__global__ __launch_bounds__(256, F::NP == 3 ? 2 : 3) void stem_wgrad_kernel(StemWArgs g) {
    constexpr int NP = F::NP;
    constexpr int PPL = 7 * WROW;                        // bytes of one patch plane (7 504)
    constexpr int APL = 64 * WAP;                        // bytes of one dY^T plane (9 216)
    constexpr int STAGE = NP * (PPL + APL);
    // fp32: ONE stage (50 KiB: three workgroups per CU, whose phases -- global loads, split + transposing stores, gathers,
    // products -- overlap each other; two stages at one workgroup per CU ran every phase of a tile back to back: 985 us);
    // 16-bit: two stages (33 KiB)
    constexpr int NST = NP == 3 ? 1 : 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, half = lane >> 5;
    typedef __attribute__((address_space(3))) unsigned char* lptr_t;
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lptr_t)lds;
    const int nkh = wave == 3 ? 1 : 2;                   // filter rows of this wave

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // ---- loaders.  Patch: 7 rows x WPW pixel slots (938) over 256 threads; dY: thread = (pixel tid & 63, channel group)
    constexpr int NSLOT = 7 * WPW, NLD = (NSLOT + 255) / 256;
    struct __attribute__((packed, aligned(4))) Rgb { float r, g, b; };
    Rgb pxs[2][NLD];                                     // two register sets: loads run TWO tiles ahead of the products
    constexpr int NDY = NP == 3 ? 4 : 2;                 // fp32: four channels per 16-byte load, sixteen groups; 16-bit: eight, eight groups
    uint4 dyrs[2][NDY];
    auto tile_of = [&](long long t64, int& img, int& oh, int& ow0) {
        const int per_img = g.Ho * g.tiles_w, t = (int)t64;      // (tiles < 2^31: checked by the host)
        img = t / per_img;
        const int rem = t - img * per_img;
        oh = rem / g.tiles_w;
        ow0 = (rem - oh * g.tiles_w) * WPX;
    };
    auto gload = [&](long long t, Rgb (&px)[NLD], uint4 (&dyr)[NDY]) {
        int img, oh, ow0;
        tile_of(t, img, oh, ow0);
        const int ih0 = 2 * oh - 3, iw0 = 2 * ow0 - 3;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx / WPW, p = idx - r * WPW;
            const int ih = ih0 + r, iw = iw0 + p;
            const bool in = idx < NSLOT && (unsigned)ih < (unsigned)g.Hin && (unsigned)iw < (unsigned)g.Win;
            const Rgb* src = reinterpret_cast<const Rgb*>(g.x + (((size_t)img * g.Hin + (in ? ih : 0)) * g.Win + (in ? iw : 0)) * 3);
            px[u] = (in && !(g.abl & 16)) ? *src : Rgb{0.f, 0.f, 0.f};
        }
        // dY: item u of a thread = (pixel 16 u + (lane & 15), channel group 4 wave + (lane >> 4)) [fp32: groups of four channels;
        // 16-bit: pixel 32 u + (lane & 31), group 2 wave + (lane >> 5) of eight] -- a load instruction reads 64 / 128-byte pieces of
        // 16 / 32 pixel rows (lane = pixel alone read 64 different cache lines per instruction: 340 of the kernel's 1 160 us), and
        // the 32 lanes of a transposing store group hit distinct banks
        const size_t row0 = ((size_t)img * g.Ho + oh) * g.Wo;
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int p = NP == 3 ? 16 * u + (lane & 15) : 32 * u + (lane & 31);
            const int grp = NP == 3 ? 4 * wave + (lane >> 4) : 2 * wave + (lane >> 5);
            const bool live = ow0 + p < g.Wo;
            const size_t row = row0 + (live ? ow0 + p : 0);
            const uint4* src = NP == 3 ? reinterpret_cast<const uint4*>(static_cast<const float*>(g.dy) + row * 64 + 4 * grp)
                                       : reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(g.dy) + row * 64 + 8 * grp);
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            if (live && !(g.abl & 1)) {
                if (g.abl & 32) {                         // (A/B: every dY row is read exactly once -- with the non-temporal hint)
                    const u32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(src));
                    dyr[u] = make_uint4(t[0], t[1], t[2], t[3]);
                } else dyr[u] = *src;
            } else dyr[u] = make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto stage_store = [&](int st, const Rgb (&px)[NLD], const uint4 (&dyr)[NDY]) {
        unsigned char* base = lds + st * STAGE;
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int idx = tid + 256 * u;
            if (idx < NSLOT) {
                unsigned char* d = base + idx * 8;
                if constexpr (NP == 3) {
                    unsigned h[2], m[2], l[2];
                    split3_pk(px[u].r, px[u].g, h[0], m[0], l[0]);
                    split3_pk(px[u].b, 0.f, h[1], m[1], l[1]);
                    *reinterpret_cast<uint2*>(d) = make_uint2(h[0], h[1]);
                    *reinterpret_cast<uint2*>(d + PPL) = make_uint2(m[0], m[1]);
                    *reinterpret_cast<uint2*>(d + 2 * PPL) = make_uint2(l[0], l[1]);
                } else {
                    *reinterpret_cast<uint2*>(d) = make_uint2(F::pack2(px[u].r, px[u].g), F::pack2(px[u].b, 0.f));
                }
            }
        }
        unsigned char* ab = base + NP * PPL;
        if (g.abl & 2) return;
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int p = NP == 3 ? 16 * u + (lane & 15) : 32 * u + (lane & 31);
            const int grp = NP == 3 ? 4 * wave + (lane >> 4) : 2 * wave + (lane >> 5);
            if constexpr (NP == 3) {
                const unsigned w[4] = {dyr[u].x, dyr[u].y, dyr[u].z, dyr[u].w};
                unsigned h[2], m[2], l[2];
                split3_pk(__uint_as_float(w[0]), __uint_as_float(w[1]), h[0], m[0], l[0]);
                split3_pk(__uint_as_float(w[2]), __uint_as_float(w[3]), h[1], m[1], l[1]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned short* d = reinterpret_cast<unsigned short*>(ab + (4 * grp + q) * WAP + p * 2);
                    const int sh = (q & 1) * 16;
                    d[0] = (unsigned short)(h[q >> 1] >> sh);
                    d[APL / 2] = (unsigned short)(m[q >> 1] >> sh);
                    d[APL] = (unsigned short)(l[q >> 1] >> sh);
                }
            } else {
                const unsigned w[4] = {dyr[u].x, dyr[u].y, dyr[u].z, dyr[u].w};
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    *reinterpret_cast<unsigned short*>(ab + (8 * grp + q) * WAP + p * 2) = (unsigned short)(w[q >> 1] >> ((q & 1) * 16));
            }
        }
    };

    // this lane's gather base inside a patch plane: column j = lane & 31 -> 2 j bytes, k-half -> 8 output pixels = 128 bytes,
    // the wave's first filter row; fragment base of the dY^T planes: channel i, k-half
    const unsigned bbase = 2 * i + 128 * half + (2 * wave) * WROW;
    const unsigned abase = NP * PPL + i * WAP + 16 * half;

    auto products = [&](int st) {
        const unsigned sb = lds0 + st * STAGE;
        const unsigned char* sa = lds + st * STAGE + abase;
#pragma unroll
        for (int ks = 0; ks < WPX / 16; ++ks) {
            uint4 af[2][NP], bf[2][NP];
            Gather gb[2][NP];
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) gb[1][p].lo[q] = gb[1][p].hi[q] = 0u;      // (wave 3 has one filter row)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int p = 0; p < NP; ++p) af[mt][p] = *reinterpret_cast<const uint4*>(sa + p * APL + mt * 32 * WAP + ks * 32);
            // (immediates: plane p, k-step ks: p * PPL + 256 ks; the wave's second filter row: + WROW in the address)
            auto row = [&](auto kc, int b) {
                constexpr int KO = decltype(kc)::value * 256;
                const unsigned addr = sb + bbase + b * WROW;
                gather8<KO>(addr, gb[b][0]);
                if constexpr (NP == 3) {
                    gather8<PPL + KO>(addr, gb[b][1]);
                    gather8<2 * PPL + KO>(addr, gb[b][2]);
                }
            };
            auto both = [&](auto kc) { row(kc, 0); if (nkh == 2) row(kc, 1); };
            if (g.abl & 4) { }
            else if (ks == 0) both(std::integral_constant<int, 0>{});
            else if (ks == 1) both(std::integral_constant<int, 1>{});
            else if (ks == 2) both(std::integral_constant<int, 2>{});
            else both(std::integral_constant<int, 3>{});
            gather_fence<NP>(gb[0]);
            gather_fence<NP>(gb[1]);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int p = 0; p < NP; ++p) bf[b][p] = merge(gb[b][p]);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                if (b < nkh && !(g.abl & 8)) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        if constexpr (NP == 3) {
                            acc[mt][b] = F::mma(af[mt][2], bf[b][0], acc[mt][b]);
                            acc[mt][b] = F::mma(af[mt][0], bf[b][2], acc[mt][b]);
                            acc[mt][b] = F::mma(af[mt][1], bf[b][1], acc[mt][b]);
                            acc[mt][b] = F::mma(af[mt][1], bf[b][0], acc[mt][b]);
                            acc[mt][b] = F::mma(af[mt][0], bf[b][1], acc[mt][b]);
                            acc[mt][b] = F::mma(af[mt][0], bf[b][0], acc[mt][b]);
                        } else {
                            acc[mt][b] = F::mma(af[mt][0], bf[b][0], acc[mt][b]);
                        }
                    }
                }
            }
        }
    };
    // Loads run two tiles ahead: a tile's 27 KiB (dY rows: 822 MB per launch in fp32 -- the kernel's main HBM stream) have two
    // tiles' products to arrive in.  Invariant at the top of a step: the stage holds tile t, set `cur` holds the loads of
    // tile t + step, the other set is free.
    const long long step = gridDim.x;
    long long t = blockIdx.x;
    if (t < g.tiles) {
        gload(t, pxs[0], dyrs[0]);
        if (t + step < g.tiles) gload(t + step, pxs[1], dyrs[1]);
        stage_store(0, pxs[0], dyrs[0]);
    }
    __syncthreads();
    int st = 0;
    while (t < g.tiles) {
#pragma unroll
        for (int cur = 1; cur >= 0; --cur) {              // cur = the set holding tile t + step's loads: 1, then 0, then 1 ...
            if (t < g.tiles) {
                if (t + 2 * step < g.tiles) gload(t + 2 * step, pxs[cur ^ 1], dyrs[cur ^ 1]);
                products(st);
                if constexpr (NST == 1) __syncthreads();  // every wave is done reading the stage
                if (t + step < g.tiles) stage_store(st ^ (NST - 1), pxs[cur], dyrs[cur]);
                __syncthreads();
                t += step;
                st ^= (NST - 1);
            }
        }
    }
    // ---- the workgroup's slab: row n = 32 mt + (accumulator row), column 32 kh + j
    float* slab = g.slabs + (size_t)blockIdx.x * 64 * 224;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        if (b < nkh) {
            const int kh = 2 * wave + b;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(size_t)(32 * mt + mfma32_row(r, half)) * 224 + 32 * kh + i] = acc[mt][b][r];
        }
    }
}

}  // namespace
}  // namespace peclr

using namespace peclr;

extern "C" int peclr_stem_pack_bytes(int fmt) {
    return fmt == 0 ? SK * 6144 : (fmt == 1 || fmt == 2) ? SK * 2048 : PECLR_ERR_UNSUPPORTED;
}

extern "C" int peclr_stem_pack(const float* w, long long stride_n, long long stride_c, long long stride_h, long long stride_w,
                               void* planes, int fmt, peclr_stream_t stream) {
    if (!w || !planes) return PECLR_ERR_NULL;
    if (fmt < 0 || fmt > 2) return PECLR_ERR_UNSUPPORTED;
    if (!aligned16(planes)) return PECLR_ERR_ALIGN;
    hipLaunchKernelGGL(stem_pack_kernel, dim3(SK), dim3(128), 0, static_cast<hipStream_t>(stream), w, (long)stride_n, (long)stride_c,
                       (long)stride_h, (long)stride_w, static_cast<unsigned char*>(planes), fmt);
    return launch_status();
}

extern "C" int peclr_stem_workgroups(int N, int Hin, int Win) {
    if (N <= 0 || Hin < 8 || Win < 8) return PECLR_ERR_SHAPE;
    const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
    const long long n = (long long)N * ((Ho + 1) / 2) * ((Wo + TPX - 1) / TPX);
    return n > 0x3fffffffLL ? PECLR_ERR_SHAPE : (int)n;
}

extern "C" int peclr_stem_conv7x7_s2(const float* x, int N, int Hin, int Win, const void* planes, int fmt, void* y,
                                     const float* stat_shift, float* stat_partial, peclr_stream_t stream) {
    if (!x || !planes || !y) return PECLR_ERR_NULL;
    if ((stat_partial != nullptr) != (stat_shift != nullptr)) return PECLR_ERR_NULL;
    const int wgs = peclr_stem_workgroups(N, Hin, Win);
    if (wgs < 0) return wgs;
    if (fmt < 0 || fmt > 2) return PECLR_ERR_UNSUPPORTED;
    if (!aligned16(planes) || !aligned16(y) || (reinterpret_cast<uintptr_t>(x) & 3u)) return PECLR_ERR_ALIGN;
    StemArgs g;
    g.x = x; g.planes = planes; g.y = y;
    g.N = N; g.Hin = Hin; g.Win = Win;
    g.Ho = (Hin - 1) / 2 + 1; g.Wo = (Win - 1) / 2 + 1;
    g.tiles_w = (g.Wo + TPX - 1) / TPX;
    g.stat_shift = stat_shift; g.stat_partial = stat_partial;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (fmt == 0) hipLaunchKernelGGL((stem_fwd_kernel<X6>), dim3(wgs), dim3(256), 0, s, g);
    else if (fmt == 1) hipLaunchKernelGGL((stem_fwd_kernel<HB>), dim3(wgs), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((stem_fwd_kernel<HF>), dim3(wgs), dim3(256), 0, s, g);
    return launch_status();
}

// Slabs (= workgroups) of the stem's weight-gradient launch: persistent workgroups, two (fp32: 228 registers) or three per CU,
// never more than there are tiles of 64 output pixels.
extern "C" int peclr_stem_wgrad_slabs(int N, int Hin, int Win, int fmt) {
    if (N <= 0 || Hin < 8 || Win < 8 || fmt < 0 || fmt > 2) return PECLR_ERR_SHAPE;
    const int Ho = (Hin - 1) / 2 + 1, Wo = (Win - 1) / 2 + 1;
    const long long tiles = (long long)N * Ho * ((Wo + WPX - 1) / WPX);
    if (tiles >= (1ll << 31)) return PECLR_ERR_SHAPE;          // (the kernel walks the tile index space with 32-bit arithmetic)
    const long long cap = fmt == 0 ? 512 : 768;
    return (int)(tiles < cap ? tiles : cap);
}

// dW slabs [peclr_stem_wgrad_slabs(...)][64][224] (column 32 kh + 4 kw + c; kw = 7 and c = 3 columns carry no meaning) of
// y = conv2d(x, W, stride 2, padding 3): x the fp32 NHWC images, dY [N][Ho][Wo][64] fp32 (fmt 0: six products of the exactly
// split operands) or bf16 / fp16 (fmt 1 / 2: one product, the images rounded to that format as the forward rounds them).
extern "C" int peclr_stem_wgrad(const float* x, const void* dY, int N, int Hin, int Win, int fmt, float* slabs, int n_slabs,
                                peclr_stream_t stream) {
    if (!x || !dY || !slabs) return PECLR_ERR_NULL;
    const int want = peclr_stem_wgrad_slabs(N, Hin, Win, fmt);
    if (want < 0) return want;
    if (n_slabs != want) return PECLR_ERR_WORKSPACE;
    if (!aligned16(dY) || !aligned16(slabs) || (reinterpret_cast<uintptr_t>(x) & 3u)) return PECLR_ERR_ALIGN;
    StemWArgs g;
    g.x = x; g.dy = dY; g.slabs = slabs;
    g.N = N; g.Hin = Hin; g.Win = Win;
    g.Ho = (Hin - 1) / 2 + 1; g.Wo = (Win - 1) / 2 + 1;
    g.tiles_w = (g.Wo + WPX - 1) / WPX;
    g.tiles = (long long)N * g.Ho * g.tiles_w;
    static const int abl = getenv("PECLR_STEM_WGRAD_ABL") ? atoi(getenv("PECLR_STEM_WGRAD_ABL")) : 0;
    g.abl = abl;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (fmt == 0) hipLaunchKernelGGL((stem_wgrad_kernel<X6>), dim3(n_slabs), dim3(256), 0, s, g);
    else if (fmt == 1) hipLaunchKernelGGL((stem_wgrad_kernel<HB>), dim3(n_slabs), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((stem_wgrad_kernel<HF>), dim3(n_slabs), dim3(256), 0, s, g);
    return launch_status();
}
